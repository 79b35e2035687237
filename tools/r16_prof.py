"""Dev tool: cycle split of one wavefront of the 16-lane-row rollout kernel (library built with -DLQMPC_R16_PROF -DPROFBLK=<block>;
LQMPC_LIB selects it).  Prints the kernel's own stderr line per launch."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, _lib
if os.environ.get('LQMPC_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['LQMPC_LIB'])
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = synth.make_batch(cfg, mix=sys.argv[2] if len(sys.argv) > 2 else 'default')
nx, nu, N, Bsz, T = b['A'].shape[0], b['B'].shape[1], b['N'], b['Bsz'], 30
dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
dJ = torch.empty(Bsz, dtype=torch.float64, device=dev)
for rep in range(3):
    s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJ)
    torch.cuda.synchronize()
print(s.last_kernel(), dJ.sum().item())
