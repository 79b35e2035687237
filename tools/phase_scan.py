"""Dev tool: where a C3 rollout launch spends its time, from the product build alone: launches with T in {1, 2, 8, 30} on the default
batch and on the same models with x0 scaled to 1e-3 (no constrained step at all: set-up + T iteration-free steps)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import os
from lq_mpc_amd import BatchSolver, synth, _lib
if os.environ.get('LQMPC_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['LQMPC_LIB'])
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = synth.make_batch(cfg, Bsz=int(sys.argv[2]) if len(sys.argv) > 2 else None)
nx, nu, N, Bsz = b['A'].shape[0], b['B'].shape[1], b['N'], b['Bsz']
dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev)
dJ = torch.empty(Bsz, dtype=torch.float64, device=dev); dit = torch.empty(Bsz, dtype=torch.int32, device=dev)
for name, scale in (('default', 1.0), ('tiny x0', 1e-3)):
    dx0 = torch.from_numpy(b['x0'] * scale).to(dev)
    for order in (-1,):
        s.set_options(order=order)
        for T in (1, 2, 8, 30):
            for _ in range(3):
                s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJ, diters=dit)
            s.timer_begin()
            for _ in range(10):
                s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJ, diters=dit)
            ms = s.timer_end() / 10
            print('%-8s order=%2d T=%2d  %.4f ms  iters/QP %.3f' % (name, order, T, ms, dit.double().sum().item() / (Bsz * T)), flush=True)
