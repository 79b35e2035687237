"""Dev experiment: C3-shaped rollout time vs number of wide-tier instances (options.nwide).  usage: c3_nwide.py Bsz nw1 nw2 ..."""
import os, sys, time, numpy as np
import torch
torch.zeros(1, device='cuda:0')
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
K = int(sys.argv[1]); nws = [int(a) for a in sys.argv[2:]]
b = synth.make_batch(3, Bsz=K)
dev = torch.device('cuda:0')
dA, dB, dx0 = (torch.from_numpy(a).to(dev) for a in (b['A'], b['B'], b['x0']))
dJ = torch.empty(K, dtype=torch.float64, device=dev); dit = torch.empty(K, dtype=torch.int32, device=dev); dst = torch.empty(K, dtype=torch.int32, device=dev)
s = BatchSolver(0)
ref = None
s.set_options(layout=0)
for nw in nws:
    s.set_options(nwide=nw)
    ts = []
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.rollout_batch_dev(4, 2, 10, K, 30, dA.data_ptr(), dB.data_ptr(), b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0.data_ptr(), b['A_true'], b['B_true'], dJ.data_ptr(), dstatus=dst.data_ptr(), diters=dit.data_ptr())
        s.sync(); ts.append(time.perf_counter() - t0)
    J = dJ.cpu().numpy()
    if ref is None: ref = J
    print('Bsz %6d nwide %6d  %s  best %.3f ms  -> %.3e QP-steps/s   max|dJ| rel %.1e  status max %d' % (K, nw, s.last_kernel(), min(ts) * 1e3, K * 30 / min(ts), np.abs(J / ref - 1).max(), int(dst.max())))
