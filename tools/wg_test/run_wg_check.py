import ctypes, numpy as np, os, sys
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libwgtest.so'))
BS, LD = 16, 17; BLK = BS * LD
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def pack(M, nb):
    out = np.zeros((nb * (nb + 1) // 2, BS, LD))
    for ib in range(nb):
        for jb in range(ib + 1):
            out[ib * (ib + 1) // 2 + jb, :, :BS] = M[ib*BS:(ib+1)*BS, jb*BS:(jb+1)*BS]
    return out
def unpack(Pk, nb):
    n = nb * BS; M = np.zeros((n, n))
    for ib in range(nb):
        for jb in range(ib + 1):
            M[ib*BS:(ib+1)*BS, jb*BS:(jb+1)*BS] = Pk[ib * (ib + 1) // 2 + jb, :, :BS]
    return M
rng = np.random.default_rng(0)
x = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), 1 << 16)); sd = np.zeros_like(x); one = np.zeros_like(x); full = np.zeros_like(x)
L.run_rsq(P(x), P(sd), P(one), P(full), x.size)
ref = 1 / np.sqrt(x.astype(np.longdouble))
print('v_rsq_f64 seed max rel err %.3e (2^%.1f); one cubic step %.3e; frsqrt %.3e' % tuple(
    [float(np.abs(sd / ref - 1).max()), float(np.log2(np.abs(sd / ref - 1).max())), float(np.abs(one / ref - 1).max()), float(np.abs(full / ref - 1).max())]))
for nb, ninst, ms_ in ((1, 4, 16), (3, 4, 7), (8, 4, 11), (8, 2048, 5), (8, 2048, 16)):
    n = nb * BS
    Ks, bs = [], []
    for i in range(min(ninst, 4)):
        X = rng.standard_normal((n, n + 5)); K = X @ X.T / n + np.eye(n); Ks.append(K); bs.append(rng.standard_normal(n))
    Kp = np.stack([pack(Ks[i % 4], nb) for i in range(ninst)]); bv = np.stack([bs[i % 4] for i in range(ninst)])
    Lo = np.zeros_like(Kp); Wo = np.zeros_like(Kp); xo = np.zeros_like(bv); so = np.zeros((ninst, BS)); ok = np.zeros(ninst, dtype=np.int32); ms = ctypes.c_float()
    cyc = np.zeros(12, dtype=np.int64); W2 = np.zeros_like(Kp)
    rc = L.run_test(P(Kp), P(bv), P(Lo), P(xo), P(Wo), P(so), nb, ms_, ninst, P(ok), ctypes.byref(ms), P(cyc), P(W2))
    errL = errx = errW = errs = 0
    for i in range(min(ninst, 4)):
        Lg = np.tril(unpack(Lo[i], nb)); Lr = np.linalg.cholesky(Ks[i]); errL = max(errL, np.abs(Lg - Lr).max())
        errx = max(errx, np.abs(xo[i] - np.linalg.solve(Ks[i], bs[i])).max())
        Wg = unpack(Wo[i], nb); Wg = np.tril(Wg) + np.tril(Wg, -1).T; errW = max(errW, np.abs(Wg - np.linalg.inv(Ks[i])).max())
        # diagonal blocks must be stored full
        Wfull = unpack(Wo[i], nb); d = max(np.abs(Wfull[k*BS:(k+1)*BS, k*BS:(k+1)*BS] - Wfull[k*BS:(k+1)*BS, k*BS:(k+1)*BS].T).max() for k in range(nb))
        errW = max(errW, d)
        errs = max(errs, np.abs(so[i, :ms_] - np.linalg.solve(Ks[i][:ms_, :ms_], bs[i][:ms_])).max())
    errW2 = max(np.abs(np.tril(unpack(W2[i], nb)) - np.tril(np.linalg.inv(Ks[i]))).max() for i in range(min(ninst, 4)))
    print(f'nb={nb} n={n} inst={ninst} m={ms_}: rc={rc} ok={ok.min()} |L| {errL:.1e} |x| {errx:.1e} |W| {errW:.1e} |small| {errs:.1e}  {ms.value:.3f} ms; ticks small {cyc[0]} chol {cyc[1]} solve {cyc[2]} triinv {cyc[3]} ztz {cyc[4]} [entry {cyc[5]} sums {cyc[6]} store {cyc[7]}]; fused chol+inverse {cyc[8]} |W2| {errW2:.1e}')
