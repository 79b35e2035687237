import sys, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = synth.make_batch(cfg)
nx, nu, N, Bsz = b['A'].shape[0], b['B'].shape[1], b['N'], b['Bsz']
dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
dJ = torch.empty(Bsz, dtype=torch.float64, device=dev)
for rep in range(2):
    for order in (-1, 0, 1):
        s.set_options(order=order)
        for _ in range(3): s.rollout_batch_dev(nx, nu, N, Bsz, 30, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJ)
        s.timer_begin()
        for _ in range(10): s.rollout_batch_dev(nx, nu, N, Bsz, 30, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJ)
        print('cfg', cfg, 'order', order, '%.4f ms' % (s.timer_end() / 10), s.last_kernel(), flush=True)
