"""Dev tool: small-batch one-shot / max-V_N / rollout times, packed vs 16-lane-row (options.layout = 0/1)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
s = BatchSolver(0)
for cfg, Bsz in ((2, 1000), (3, 1000), (3, 4096), (3, 16384)):
    b = synth.make_batch(cfg, Bsz=Bsz)
    a = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
    x0s = np.ascontiguousarray(b['x0'][:, :8])
    for env in ('0', '1'):
        s.set_options(layout=int(env))
        res = {}
        for name, fn in (('solve', lambda: s.solve_batch(*a, b['x0'])), ('maxvn', lambda: s.max_vn_batch(*a, x0s)), ('rollout', lambda: s.rollout_batch(30, *a, b['x0'], b['A_true'], b['B_true']))):
            ts = []
            for rep in range(5):
                t0 = time.perf_counter(); g = fn(); ts.append(time.perf_counter() - t0)
            res[name] = (min(ts) * 1e3, s.last_kernel())
        print('C%d Bsz %5d layout=%s: ' % (cfg, Bsz, env) + '  '.join('%s %.3f ms (%s)' % (k, v[0], v[1].split('<')[0].replace('lqmpc_', '')) for k, v in res.items()) + '   [host-inclusive]')
