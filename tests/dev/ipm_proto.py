"""Dev tool (not product, not oracle): numpy batched prototype of the kernel's IPM, for tuning."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import synth
from oracle import oracle as orc

def condense_batch(A, B, Q, R, P, N):
    nx, nu, Bsz = B.shape
    n = N * nu
    H = np.zeros((Bsz, n, n)); F = np.zeros((Bsz, n, nx))
    for b in range(Bsz):
        H[b], F[b] = orc.condense(A[:, :, b], B[:, :, b], Q, R, P, N)
    return H, F

def ipm(Pm, q, h, tol=1e-9, maxit=40, z0_mode='q', polish=False, tau=0.995):
    """min 1/2 v'Pv + q'v, |v|<=h. Pm (B,n,n), q (B,n). Mehrotra PC, primal feasible, slacks tracked."""
    Bsz, n = q.shape
    v = np.zeros((Bsz, n))
    sl = np.full((Bsz, n), h); su = np.full((Bsz, n), h)
    qs = np.max(np.abs(q), axis=1, keepdims=True)
    if z0_mode == 'q':
        z0 = np.maximum(qs, 1e-3) * np.ones((1, n))
    else:
        z0 = np.full((Bsz, n), float(z0_mode))
    zl = z0.copy(); zu = z0.copy()
    iters = np.zeros(Bsz, dtype=int)
    done = np.zeros(Bsz, dtype=bool)
    scale = np.maximum(qs[:, 0], 1e-3)
    rd = np.einsum('bij,bj->bi', Pm, v) + q - zl + zu
    for it in range(maxit):
        mu = (np.sum(sl * zl, 1) + np.sum(su * zu, 1)) / (2 * n)
        conv = (mu <= tol * scale * h) & (np.max(np.abs(rd), 1) <= tol * scale)
        done |= conv
        if done.all(): break
        iters[~done] += 1
        isl = 1 / sl; isu = 1 / su
        d = zl * isl + zu * isu
        K = Pm + np.einsum('bi,ij->bij', d, np.eye(n))
        ksolve = lambda r: np.linalg.solve(K, r[..., None])[..., 0]
        dva = ksolve(-rd - zl + zu)
        dzla = -zl - zl * isl * dva
        dzua = -zu + zu * isu * dva
        def steplen(x, dx, tau):
            r = np.where(dx < 0, -dx / x, 0.0)
            m = np.max(r, axis=1)
            return np.where(m > tau, tau / np.maximum(m, 1e-300), 1.0)
        ap = np.minimum(steplen(sl, dva, 1.0), steplen(su, -dva, 1.0))
        ad = np.minimum(steplen(zl, dzla, 1.0), steplen(zu, dzua, 1.0))
        mua = (np.sum((sl + ap[:, None] * dva) * (zl + ad[:, None] * dzla), 1) +
               np.sum((su - ap[:, None] * dva) * (zu + ad[:, None] * dzua), 1)) / (2 * n)
        sig = (mua / mu) ** 3
        rcl = (sig * mu)[:, None] - sl * zl - dva * dzla
        rcu = (sig * mu)[:, None] - su * zu + dva * dzua
        rhs = -rd + rcl * isl - rcu * isu
        dv = ksolve(rhs)
        dzl = (rcl - zl * dv) * isl
        dzu = (rcu + zu * dv) * isu
        ap = np.minimum(steplen(sl, dv, tau), steplen(su, -dv, tau))
        ad = np.minimum(steplen(zl, dzl, tau), steplen(zu, dzu, tau))
        ap = np.where(done, 0.0, ap)[:, None]; ad = np.where(done, 0.0, ad)[:, None]
        v = v + ap * dv; sl = sl + ap * dv; su = su - ap * dv
        zl = zl + ad * dzl; zu = zu + ad * dzu
        # tracked residual: rd_new = (1-ap) rd + (ap-ad)(dzl-dzu)
        rd = (1 - ap) * rd + (ap - ad) * (dzl - dzu)
    rd_true = np.einsum('bij,bj->bi', Pm, v) + q - zl + zu
    drift = np.max(np.abs(rd_true - rd))
    if polish:
        actl = zl > sl; actu = zu > su
        act = actl | actu
        vb = np.where(actl, -h, np.where(actu, h, 0.0))
        # masked system: rows/cols of active replaced by identity
        free = ~act
        Km = Pm * (free[:, :, None] & free[:, None, :]) + np.einsum('bi,ij->bij', act.astype(float), np.eye(n))
        rhs = np.where(act, vb, -(q + np.einsum('bij,bj->bi', Pm, vb)))
        vp = np.linalg.solve(Km, rhs[..., None])[..., 0]
        grad = np.einsum('bij,bj->bi', Pm, vp) + q
        ok = (np.abs(vp) <= h * (1 + 1e-12)).all(1) & (np.where(actl, grad >= -1e-9, True)).all(1) & (np.where(actu, grad <= 1e-9, True)).all(1)
        v = np.where(ok[:, None], vp, v)
        return v, iters, done, drift, ok
    return v, iters, done, drift, None

if __name__ == '__main__':
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    Bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    b = synth.make_batch(cfg, Bsz=Bsz)
    H, F = condense_batch(b['A'], b['B'], b['Q'], b['R'], b['P'], b['N'])
    x0 = b['x0'].T
    g = np.einsum('bij,bj->bi', F, x0)
    ref = orc.solve_batch(b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'], b['x0'])
    for polish in (False, True):
      for tol in (1e-6, 1e-8, 1e-10, 1e-12, 1e-14):
        for z0 in ('q', 1.0):
            v, iters, done, drift, ok = ipm(2 * H, 2 * g, 0.1, tol=tol, z0_mode=z0, polish=polish)
            u0 = v[:, :b['nu']].T
            err = np.max(np.abs(u0 - ref['u_0']) / np.maximum(np.abs(ref['u_0']), 1e-4))
            print(f"polish={polish} tol={tol:g} z0={z0}: iters mean {iters.mean():.2f} max {iters.max()} done {done.mean():.3f} u0 relerr {err:.2e} drift {drift:.1e} ok {None if ok is None else ok.mean()}")
    print('AS iters mean', ref['iters'].mean(), 'frac with active', np.mean(np.abs(np.abs(ref['u_0']).max(0) - 0.1) < 1e-12))

def ipm_ws(Pm, q, h, tol=1e-12, maxit=40, delta=0.2, mu0=1e-2, tau=0.995, sig_pow=3):
    """warm start from clipped unconstrained minimiser"""
    Bsz, n = q.shape
    vunc = np.linalg.solve(Pm, -q[..., None])[..., 0]
    inside = (np.abs(vunc) <= h).all(1)
    v = np.clip(vunc, -(1 - delta) * h, (1 - delta) * h)
    sl = h + v; su = h - v
    gr = np.einsum('bij,bj->bi', Pm, v) + q
    qs = np.max(np.abs(q), axis=1, keepdims=True)
    scale = np.maximum(qs[:, 0], 1e-3)
    m0 = mu0 * scale[:, None] * h
    zl = m0 / sl + np.maximum(gr, 0); zu = m0 / su + np.maximum(-gr, 0)
    iters = np.zeros(Bsz, dtype=int)
    done = inside.copy()
    rd = gr - zl + zu
    for it in range(maxit):
        mu = (np.sum(sl * zl, 1) + np.sum(su * zu, 1)) / (2 * n)
        conv = (mu <= tol * scale * h) & (np.max(np.abs(rd), 1) <= tol * scale)
        done |= conv
        if done.all(): break
        iters[~done] += 1
        isl = 1 / sl; isu = 1 / su
        d = zl * isl + zu * isu
        K = Pm + np.einsum('bi,ij->bij', d, np.eye(n))
        ksolve = lambda r: np.linalg.solve(K, r[..., None])[..., 0]
        dva = ksolve(-rd - zl + zu)
        dzla = -zl - zl * isl * dva
        dzua = -zu + zu * isu * dva
        def steplen(x, dx, tau):
            r = np.where(dx < 0, -dx / x, 0.0)
            m = np.max(r, axis=1)
            return np.where(m > tau, tau / np.maximum(m, 1e-300), 1.0)
        ap = np.minimum(steplen(sl, dva, 1.0), steplen(su, -dva, 1.0))
        ad = np.minimum(steplen(zl, dzla, 1.0), steplen(zu, dzua, 1.0))
        mua = (np.sum((sl + ap[:, None] * dva) * (zl + ad[:, None] * dzla), 1) +
               np.sum((su - ap[:, None] * dva) * (zu + ad[:, None] * dzua), 1)) / (2 * n)
        sig = (mua / mu) ** sig_pow
        rcl = (sig * mu)[:, None] - sl * zl - dva * dzla
        rcu = (sig * mu)[:, None] - su * zu + dva * dzua
        rhs = -rd + rcl * isl - rcu * isu
        dv = ksolve(rhs)
        dzl = (rcl - zl * dv) * isl
        dzu = (rcu + zu * dv) * isu
        ap = np.minimum(steplen(sl, dv, tau), steplen(su, -dv, tau))
        ad = np.minimum(steplen(zl, dzl, tau), steplen(zu, dzu, tau))
        ap = np.where(done, 0.0, ap)[:, None]; ad = np.where(done, 0.0, ad)[:, None]
        v = v + ap * dv; sl = sl + ap * dv; su = su - ap * dv
        zl = zl + ad * dzl; zu = zu + ad * dzu
        rd = (1 - ap) * rd + (ap - ad) * (dzl - dzu)
    v = np.where(inside[:, None], vunc, v)
    return v, iters, done, inside

def run_ws():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    Bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    b = synth.make_batch(cfg, Bsz=Bsz)
    H, F = condense_batch(b['A'], b['B'], b['Q'], b['R'], b['P'], b['N'])
    g = np.einsum('bij,bj->bi', F, b['x0'].T)
    ref = orc.solve_batch(b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'], b['x0'])
    for delta in (0.05, 0.2, 0.5):
        for mu0 in (1e-1, 1e-2, 1e-3):
            v, iters, done, inside = ipm_ws(2 * H, 2 * g, 0.1, delta=delta, mu0=mu0)
            u0 = v[:, :b['nu']].T
            err = np.max(np.abs(u0 - ref['u_0']) / np.maximum(np.abs(ref['u_0']), 1e-4))
            act = ~inside
            print(f"delta={delta} mu0={mu0}: iters(mean over constrained) {iters[act].mean():.2f} max {iters.max()} done {done.mean():.3f} err {err:.1e} inside {inside.mean():.3f}")
if len(sys.argv) > 3 and sys.argv[3] == 'ws':
    run_ws()

def run_z0():
    cfg = int(sys.argv[1]); Bsz = int(sys.argv[2])
    b = synth.make_batch(cfg, Bsz=Bsz)
    H, F = condense_batch(b['A'], b['B'], b['Q'], b['R'], b['P'], b['N'])
    g = np.einsum('bij,bj->bi', F, b['x0'].T)
    print('|q|inf quantiles', np.quantile(np.max(np.abs(2*g),1), [0,0.1,0.5,0.9,1]))
    print('cond(H) median/max', np.median(np.linalg.cond(H)), np.max(np.linalg.cond(H)))
    for c in (0.03, 0.1, 0.3, 1.0):
        for tau in (0.99, 0.995, 0.999):
            qs = np.max(np.abs(2*g), axis=1)
            # emulate z0 = c*|q|inf by scaling the problem: pass z0_mode numeric per-batch not supported -> rescale q and P
            s = (c * qs)[:, None]
            v, iters, done, drift, ok = ipm(2 * H / s[:, :, None], 2 * g / s, 0.1, tol=1e-12, z0_mode=1.0, tau=tau)
            print(f"c={c} tau={tau}: iters mean {iters.mean():.2f} max {iters.max()} p90 {np.quantile(iters,0.9)}")
if len(sys.argv) > 3 and sys.argv[3] == 'z0':
    run_z0()
