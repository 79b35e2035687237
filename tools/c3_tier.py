"""Dev experiment: hardest 1024 C3 instances on the LPS=4 kernel vs the whole-wave (LPS=64) kernel."""
import os, sys, time, numpy as np
import torch
torch.zeros(1, device='cuda:0')
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
b = synth.make_batch(3)
s = BatchSolver(0)
args = lambda bb: (bb['N'], bb['A'], bb['B'], bb['Q'], bb['R'], bb['P'], bb['lb'], bb['ub'])
r = s.rollout_batch(30, *args(b), b['x0'], b['A_true'], b['B_true'])
order = np.argsort(-r['iters'])
for K in (256, 1024, 4096):
    idx = order[:K]
    sub = dict(b, A=np.ascontiguousarray(b['A'][:, :, idx]), B=np.ascontiguousarray(b['B'][:, :, idx]), Bsz=K)
    x0 = np.ascontiguousarray(b['x0'][:, idx])
    dev = torch.device('cuda:0')
    dA, dB, dx0 = (torch.from_numpy(a).to(dev) for a in (sub['A'], sub['B'], x0))
    dJ = torch.empty(K, dtype=torch.float64, device=dev); dit = torch.empty(K, dtype=torch.int32, device=dev); dst = torch.empty(K, dtype=torch.int32, device=dev)
    for tag in ('lps4', 'lps64'):
        if tag == 'lps64': os.environ['LQMPC_EXP_LPS64'] = '1'
        else: os.environ.pop('LQMPC_EXP_LPS64', None)
        s.set_options(order=0)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.rollout_batch_dev(4, 2, 10, K, 30, dA.data_ptr(), dB.data_ptr(), b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0.data_ptr(), b['A_true'], b['B_true'], dJ.data_ptr(), dstatus=dst.data_ptr(), diters=dit.data_ptr())
            s.sync(); ts.append(time.perf_counter() - t0)
        print(K, tag, s.last_kernel(), 'best %.3f ms' % (min(ts) * 1e3), 'iters mean', dit.float().mean().item(), 'J sum', dJ.sum().item())
os.environ.pop('LQMPC_EXP_LPS64', None)
