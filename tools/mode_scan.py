"""Dev tool: device-resident one-shot / max-V_N / sweep times over batch sizes, packed vs 16-lane-row (options.layout = 0/1)."""
import os, sys, time, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, _lib
if os.environ.get('LQMPC_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['LQMPC_LIB'])
from lq_mpc_amd.mpc import _ptr
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sizes = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1024, 8192, 16384, 65536]
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
T, K = 30, 8
for Bsz in sizes:
    b = synth.make_batch(cfg, Bsz=Bsz)
    nx, nu, N = b['A'].shape[0], b['B'].shape[1], b['N']
    dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
    x0s = np.ascontiguousarray(b['x0'][:, :K])
    du0 = torch.empty((nu, Bsz), dtype=torch.float64, device=dev)
    dV = torch.empty(Bsz, dtype=torch.float64, device=dev); dJ = torch.empty(Bsz, dtype=torch.float64, device=dev)
    dit = torch.empty(Bsz, dtype=torch.int32, device=dev); dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
    c = (b['Q'], b['R'], b['P'], b['lb'], b['ub'])
    def sweep():
        _lib.check(s._L.lqmpc_sweep_batch_dev(s._h, nx, nu, N, Bsz, T, K, _ptr(dA), _ptr(dB), *[_ptr(v) for v in c], _ptr(dx0), _ptr(x0s),
                                              _ptr(b['A_true']), _ptr(b['B_true']), 0, None, None, _ptr(dJ), _ptr(dV), _ptr(dst), _ptr(dit)))
    fns = (('solve', lambda: s.solve_batch_dev(nx, nu, N, Bsz, dA, dB, *c, dx0, du0, dV, dstatus=dst, diters=dit), lambda: dV.sum().item()),
           ('maxvn', lambda: s.max_vn_batch_dev(nx, nu, N, Bsz, dA, dB, *c, x0s, dV, dstatus=dst, diters=dit), lambda: dV.sum().item()),
           ('sweep', sweep, lambda: dV.sum().item() + dJ.sum().item()))
    for env in ('0', '1'):
        s.set_options(layout=int(env))
        out = []
        for name, fn, chk in fns:
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(5): fn()
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 5)
            out.append('%s %.3f ms [%s bad %d sum %.9g]' % (name, best * 1e3, s.last_kernel().split('<')[0].replace('lqmpc_', ''), int((dst != 0).sum()), chk()))
        print('C%d Bsz %6d layout=%s: ' % (cfg, Bsz, env) + '  '.join(out), flush=True)
