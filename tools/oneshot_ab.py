import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
b = synth.make_batch(3)
dev = torch.device('cuda', 0)
nx, nu, N, Bsz = 4, 2, 10, b['Bsz']
dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
du0 = torch.empty((nu, Bsz), dtype=torch.float64, device=dev); dVN = torch.empty(Bsz, dtype=torch.float64, device=dev)
dit = torch.empty(Bsz, dtype=torch.int32, device=dev); dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
for name, kw in (('ipm+polish', dict(presolve=0, warm_start=0)), ('presolve+warm', dict(presolve=1, warm_start=1)), ('warm only', dict(presolve=0, warm_start=1))):
    s.set_options(**kw)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20):
            s.solve_batch_dev(nx, nu, N, Bsz, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, du0, dVN, dstatus=dst, diters=dit)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f'{name:14s} {dt*1e3:7.3f} ms  {Bsz/dt/1e6:8.1f} M QP/s  fact_mean {dit.double().mean().item():.3f} bad {int((dst!=0).sum())} checksum {dVN.sum().item():.10f}')
