"""Dev tool: per-call wall time of the host-buffer entry points (numpy in/out) over batch sizes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, LQ_MPC_Controller
s = BatchSolver(0)
b = synth.make_batch(3, Bsz=8192)
sh = (b['Q'], b['R'], b['P'], b['lb'], b['ub'])
for m in (1, 16, 64, 256, 1024, 4096, 8192):
    A, B, x = b['A'][:, :, :m].copy(), b['B'][:, :, :m].copy(), b['x0'][:, :m].copy()
    for name, fn in (('solve', lambda: s.solve_batch(b['N'], A, B, *sh, x)), ('rollout30', lambda: s.rollout_batch(30, b['N'], A, B, *sh, x, b['A_true'], b['B_true']))):
        for _ in range(5): fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20): fn()
            ts.append((time.perf_counter() - t0) / 20)
        print('m=%5d %-9s median %.1f us  min %.1f us' % (m, name, np.median(ts) * 1e6, min(ts) * 1e6), flush=True)
c = LQ_MPC_Controller(10, b['A'][:, :, 0], b['B'][:, :, 0], b['Q'], b['R'], b['P'], np.vstack((10 * np.eye(2), -10 * np.eye(2))), solver=s)
x0 = b['x0'][:, 0].copy(); z = np.zeros((4, 10)); zu = np.zeros((2, 10))
for _ in range(5): c.solve(x0, z, zu)
t0 = time.perf_counter()
for _ in range(200): c.solve(x0, z, zu)
print('LQ_MPC_Controller.solve: %.1f us per call' % ((time.perf_counter() - t0) / 200 * 1e6))
