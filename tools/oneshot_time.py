"""Dev tool: launch time of the one-shot solve (lqmpc_solve_batch_dev, default options) of a config in the default and the hard mix."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, _lib
if os.environ.get('LQMPC_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['LQMPC_LIB'])
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
for cfg in [int(a) for a in sys.argv[1:]] or [3]:
    for mix in ('default', 'hard'):
        b = synth.make_batch(cfg, mix=mix)
        nx, nu, N, Bsz = b['A'].shape[0], b['B'].shape[1], b['N'], b['Bsz']
        dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
        du0 = torch.empty((nu, Bsz), dtype=torch.float64, device=dev); dVN = torch.empty(Bsz, dtype=torch.float64, device=dev)
        dit = torch.empty(Bsz, dtype=torch.int32, device=dev); dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
        go = lambda: s.solve_batch_dev(nx, nu, N, Bsz, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, du0, dVN, dstatus=dst, diters=dit)
        for _ in range(3): go()
        s.timer_begin()
        for _ in range(10): go()
        ms = s.timer_end() / 10
        print(f"C{cfg} {mix:8s} {ms:8.4f} ms  {Bsz / ms * 1e3:.3e} QPs/s  iters/QP {dit.double().mean().item():.3f}  status!=0 {int((dst != 0).sum())}  {s.last_kernel()}", flush=True)
