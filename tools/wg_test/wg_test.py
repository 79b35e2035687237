import ctypes, numpy as np, os, sys
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libwgtest.so'))
BS, LD = 16, 17; BLK = BS * LD
def pack(M, nb):
    out = np.zeros((nb * (nb + 1) // 2, BS, LD))
    for ib in range(nb):
        for jb in range(ib + 1):
            out[ib * (ib + 1) // 2 + jb, :, :BS] = M[ib*BS:(ib+1)*BS, jb*BS:(jb+1)*BS]
    return out
def unpack(P, nb):
    n = nb * BS; M = np.zeros((n, n))
    for ib in range(nb):
        for jb in range(ib + 1):
            M[ib*BS:(ib+1)*BS, jb*BS:(jb+1)*BS] = P[ib * (ib + 1) // 2 + jb, :, :BS]
    return M
rng = np.random.default_rng(0)
for nb, ninst in ((1, 4), (3, 4), (8, 4), (8, 2048)):
    n = nb * BS
    Ks, bs = [], []
    for i in range(min(ninst, 4)):
        X = rng.standard_normal((n, n + 5)); K = X @ X.T / n + np.eye(n); Ks.append(K); bs.append(rng.standard_normal(n))
    Kp = np.stack([pack(Ks[i % 4], nb) for i in range(ninst)]); bv = np.stack([bs[i % 4] for i in range(ninst)])
    Lo = np.zeros_like(Kp); xo = np.zeros_like(bv); ok = np.zeros(ninst, dtype=np.int32); ms = ctypes.c_float()
    rc = L.run_test(Kp.ctypes.data_as(ctypes.c_void_p), bv.ctypes.data_as(ctypes.c_void_p), Lo.ctypes.data_as(ctypes.c_void_p), xo.ctypes.data_as(ctypes.c_void_p), nb, ninst, ok.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ms))
    errL = errx = 0
    for i in range(min(ninst, 4)):
        Lg = np.tril(unpack(Lo[i], nb)); Lr = np.linalg.cholesky(Ks[i]); errL = max(errL, np.abs(Lg - Lr).max())
        errx = max(errx, np.abs(xo[i] - np.linalg.solve(Ks[i], bs[i])).max())
    print(f'nb={nb} n={n} inst={ninst}: rc={rc} ok={ok.min()} max|L-Lref|={errL:.2e} max|x-xref|={errx:.2e} time {ms.value:.3f} ms -> {ninst/ms.value/1e3:.3f} M fact+solve/s')
