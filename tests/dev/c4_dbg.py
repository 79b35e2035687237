"""Dev: per-instance errors of the (4,2,20) random problems of tests/test_gpu_parity.py (one-shot and rollout) against the oracle."""
import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver
from oracle import oracle as orc
nx, nu, N, warm = 4, 2, 20, 1
rng = np.random.default_rng(100 * nx + N + warm)
Bsz, T = 768, 12
A = rng.standard_normal((nx, nx, Bsz))
rho_hi = 1.3
A *= rng.uniform(0.3, rho_hi, Bsz) / np.abs(np.linalg.eigvals(A.transpose(2, 0, 1))).max(axis=1)
B = rng.standard_normal((nx, nu, Bsz)) * rng.uniform(0.1, 2.0, (1, 1, Bsz))
def spd(m, lo, hi):
    M = rng.standard_normal((m, m)); M = M @ M.T / m + np.eye(m)
    return M * rng.uniform(lo, hi)
Q, R, P = spd(nx, 0.5, 5.0), spd(nu, 0.05, 2.0), spd(nx, 0.5, 20.0)
lb, ub = -rng.uniform(0.05, 0.5, nu), rng.uniform(0.05, 0.5, nu)
x0 = rng.standard_normal((nx, Bsz)) * rng.choice([1e-3, 0.1, 1.0, 10.0], Bsz)
xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.05 * rng.standard_normal((nu, N))
A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
s = BatchSolver(0)
for refs in (True, False):
    a = (xr, ur) if refs else ()
    g1 = s.solve_batch(N, A, B, Q, R, P, lb, ub, x0, *a)
    r1 = orc.solve_batch(N, A, B, Q, R, P, lb, ub, x0, *a)
    e = np.abs(g1["V_N"] - r1["V_N"]) / np.maximum(1.0, np.abs(r1["V_N"]))
    bad = np.where(e > 1e-7)[0]
    print('refs', refs, s.last_kernel(), 'bad', len(bad), 'of', Bsz, 'max', e.max(), 'status', np.unique(g1["status"]), 'iters bad', g1["iters"][bad][:10], 'iters median', np.median(g1["iters"]))
    print('   bad idx', bad[:16], 'e', e[bad][:8])
