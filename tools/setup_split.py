"""Dev tool: how the packed kernel's per-instance setup splits (one-shot solves, C3)."""
import sys, time, numpy as np
import torch
torch.zeros(1, device='cuda:0')
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
b = synth.make_batch(3); K = b['Bsz']; dev = torch.device('cuda:0')
dA, dB, dx0 = (torch.from_numpy(a).to(dev) for a in (b['A'], b['B'], b['x0']))
dz = torch.zeros_like(dx0)
du = torch.empty((2, K), dtype=torch.float64, device=dev); dV = torch.empty(K, dtype=torch.float64, device=dev)
dit = torch.empty(K, dtype=torch.int32, device=dev); dst = torch.empty(K, dtype=torch.int32, device=dev)
s = BatchSolver(0)
def run(tag, x, **opt):
    s.set_options(**opt)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.solve_batch_dev(4, 2, 10, K, dA.data_ptr(), dB.data_ptr(), b['Q'], b['R'], b['P'], b['lb'], b['ub'], x.data_ptr(), du.data_ptr(), dV.data_ptr(), dstatus=dst.data_ptr(), diters=dit.data_ptr())
        s.sync(); ts.append(time.perf_counter() - t0)
    print('%-48s %.3f ms  iters mean %.2f' % (tag, min(ts) * 1e3, dit.float().mean().item()))
run('x0 = 0, presolve on  (condense + chol + G, 0 it)', dz, presolve=1, warm_start=1)
run('x0 = 0, presolve off (condense + IPM from 0)', dz, presolve=0, warm_start=0, polish=0, max_iter=1)
run('default one-shot', dx0, presolve=-1, warm_start=-1, polish=1, max_iter=50)
