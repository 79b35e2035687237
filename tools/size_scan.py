import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
dev = torch.device('cuda', 0)
nx, nu, N, T = 4, 2, 10, 30
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
full = synth.make_batch(3)
for Bsz in (1024, 4096, 16384, 32768, 65536, 131072):
    b = synth.make_batch(3, Bsz=Bsz)
    dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
    dJT = torch.empty(Bsz, dtype=torch.float64, device=dev); dit = torch.empty(Bsz, dtype=torch.int32, device=dev); dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
    for order in (0, 1):
        s.set_options(order=order)
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5):
                s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJT, dstatus=dst, diters=dit)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f'Bsz {Bsz:7d} order {order}: {dt*1e3:7.3f} ms  {Bsz*T/dt/1e6:8.1f} M QP-steps/s  max fact/instance {int(dit.max())}')
