import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
b = synth.make_batch(3)
dev = torch.device('cuda', 0)
nx, nu, N, Bsz, T = 4, 2, 10, b['Bsz'], 30
dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
dJT = torch.empty(Bsz, dtype=torch.float64, device=dev); dit = torch.empty(Bsz, dtype=torch.int32, device=dev); dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
import os
for name, kw in (('natural warm', dict(order=0, warm_start=1)), ('sorted warm key=' + os.environ.get('LQMPC_KEY_MODE', '0'), dict(order=1, warm_start=1))):
    s.set_options(**kw)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJT, dstatus=dst, diters=dit)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f'{name:12s} {dt*1e3:7.3f} ms  iters_mean {dit.double().mean().item()/T:.3f} checksum {dJT.sum().item():.12f}')
