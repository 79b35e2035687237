"""Dev experiment (CPU): iterations of the primal-dual active-set method at the cold start (step 0 of a C3 rollout) from different
first guesses of the active set: (a) the rows of the unconstrained minimiser that violate the box (what the kernels use),
(b) the stages where a saturated time-varying LQR roll-forward clips."""
import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import synth

def pdas(H, g, h, L0, U0, maxit=30, full=False):
    n = len(g)
    L, U = L0.copy(), U0.copy()
    for it in range(1, maxit + 1):
        Aset = L | U
        F = ~Aset
        v = np.zeros(n)
        v[L] = -h[L]; v[U] = h[U]
        if F.any():
            v[F] = np.linalg.solve(H[np.ix_(F, F)], -(g[F] + H[np.ix_(F, Aset)] @ v[Aset]))
        lam = H @ v + g                      # gradient: must be >= 0 at lower bounds, <= 0 at upper bounds
        nL = (F & (v < -h * (1 + 1e-12))) | (L & (lam >= 0))
        nU = (F & (v > h * (1 + 1e-12))) | (U & (lam <= 0))
        if (nL == L).all() and (nU == U).all():
            return (it, v, L, U) if full else (it, v)
        L, U = nL, nU
    return (maxit, v, L, U) if full else (maxit, v)

if __name__ == '__main__':
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    mix = sys.argv[3] if len(sys.argv) > 3 else 'default'
    b = synth.make_batch(cfg, Bsz=K, mix=mix)
    nx, nu, N = b['nx'], b['nu'], b['N']
    n = N * nu
    h = np.tile(0.5 * (b['ub'] - b['lb']), N)
    Q, R, P = b['Q'], b['R'], b['P']
    ia, ib, busy = [], [], 0
    for k in range(K):
        A, B, x0 = b['A'][:, :, k], b['B'][:, :, k], b['x0'][:, k]
        H, F = synth.condense_np(A, B, Q, R, P, N)[:2]
        g = F @ x0 if F.shape[1] == nx else F.T @ x0
        vunc = np.linalg.solve(H, -g)
        L0, U0 = vunc < -h, vunc > h
        if not (L0 | U0).any(): continue
        busy += 1
        it_a, va = pdas(H, g, h, L0, U0)
        # saturated time-varying LQR roll-forward
        S = P.copy(); Ks = [None] * N
        for j in range(N - 1, -1, -1):
            Re = R + B.T @ S @ B
            Kj = np.linalg.solve(Re, B.T @ S @ A)
            S = Q + A.T @ S @ (A - B @ Kj)
            Ks[j] = Kj
        x = x0.copy(); Lb = np.zeros(n, bool); Ub = np.zeros(n, bool)
        for j in range(N):
            u = -Ks[j] @ x
            Lb[j * nu:(j + 1) * nu] = u < -h[:nu]; Ub[j * nu:(j + 1) * nu] = u > h[:nu]
            u = np.clip(u, -h[:nu], h[:nu])
            x = A @ x + B @ u
        it_b, vb = pdas(H, g, h, Lb, Ub)
        assert np.allclose(va, vb, atol=1e-8), (k, np.abs(va - vb).max())
        ia.append(it_a); ib.append(it_b)
    ia, ib = np.array(ia), np.array(ib)
    print('C%d %s: %d of %d instances constrained at step 0' % (cfg, mix, busy, K))
    print('  violated rows of v_unc : mean %.2f iterations, histogram %s' % (ia.mean(), np.bincount(ia)[:10]))
    print('  saturated LQR roll     : mean %.2f iterations, histogram %s' % (ib.mean(), np.bincount(ib)[:10]))
