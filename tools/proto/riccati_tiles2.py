"""Dev prototype (numpy, CPU), second version: the matrix-core set-up with state matrices of up to 8 x 8 (2 x 2 tiles of 4 x 4) and
stage blocks of 1..4 inputs (3 is padded to 4 with a dummy input: zero column of B, unit weight), as lqmpc_r16_setup.h does it.
Every product is D = A'B + C on 4 x 4 tiles; the inverse of Re is built from products and element-wise reciprocals only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from lq_mpc_amd import synth

def mm(a, b, c=None):
    d = a.T @ b
    return d if c is None else d + c

Z4 = lambda: np.zeros((4, 4))

def inv2(M, o):
    """inverse of the 2x2 block of M at rows/cols o..o+1 (zero elsewhere), from products: adj = J M J', det I = M adj"""
    Jt = Z4(); Jt[o, o + 1] = -1.0; Jt[o + 1, o] = 1.0
    ones = Z4(); ones[o:o + 2, o:o + 2] = 1.0
    blk = Z4(); blk[o:o + 2, o:o + 2] = M[o:o + 2, o:o + 2]
    adj = mm(mm(blk, Jt), Jt)
    det = mm(ones, mm(blk, adj))
    out = Z4(); out[o:o + 2, o:o + 2] = adj[o:o + 2, o:o + 2] / det[o:o + 2, o:o + 2]
    return out

def small_inverse(Re, nup):
    if nup == 1:
        out = Z4(); out[0, 0] = 1.0 / Re[0, 0]; return out
    if nup == 2:
        return inv2(Re, 0)
    # 4 x 4 by 2 x 2 blocks: [E F; F' H]^-1 with S = H - F'E^-1 F
    mTR = Z4(); mTR[0:2, 2:4] = 1.0
    mBR = Z4(); mBR[2:4, 2:4] = 1.0
    Ei = inv2(Re, 0)
    Fb = Re * mTR
    EiF = mm(Ei, Fb)                      # E^-1 F   (rows 0-1, cols 2-3)
    S = Re * mBR - mm(Fb, EiF)            # H - F' E^-1 F
    Si = inv2(S, 2)
    FtEi = mm(Fb, Ei)                     # F' E^-1  (rows 2-3, cols 0-1)
    X = mm(FtEi, Si)                      # E^-1 F S^-1
    Xt = mm(Si, FtEi)                     # S^-1 F' E^-1
    return Ei + mm(Xt, FtEi) - X - Xt + Si

def setup(A_, B_, Q_, R_, PT_, N):
    nx, nu = B_.shape
    TX = (nx + 3) // 4
    nup = 4 if nu == 3 else nu
    SPT = 4 // nup
    npad = N * nup; NT = (npad + 3) // 4
    def sq(M):
        P = np.zeros((4 * TX, 4 * TX)); P[:M.shape[0], :M.shape[1]] = M
        return [[P[4 * a:4 * a + 4, 4 * b:4 * b + 4].copy() for b in range(TX)] for a in range(TX)]
    def col(M, c0=0):           # nx x (<= 4) matrix placed at columns c0.. of an NX x 4 column of tiles
        P = np.zeros((4 * TX, 4)); P[:M.shape[0], c0:c0 + M.shape[1]] = M
        return [P[4 * a:4 * a + 4].copy() for a in range(TX)]
    A = sq(A_); At = sq(A_.T); Q = sq(Q_); S = sq(PT_)
    Bp = col(B_); Bt = [t.T.copy() for t in Bp]          # Bt[a] = (Bp[a])' as a register matrix
    nBt = [-t for t in Bt]
    Bpl = [col(B_, q * nup) for q in range(SPT)]; nBpl = [[-t for t in c] for c in Bpl]
    Rp = Z4(); Rp[:nu, :nu] = R_
    for u in range(nu, nup): Rp[u, u] = 1.0
    rho = [[Z4() for _ in range(TX)] for _ in range(NT)]
    W = {(I, J): Z4() for I in range(NT) for J in range(I + 1)}
    TtA = [Z4() for _ in range(NT)]; TDA = [Z4() for _ in range(NT)]
    R = range(TX)
    for j in range(N - 1, -1, -1):
        Ij, off = (j * nup) // 4, (j * nup) % 4
        q = off // nup
        first = (j == N - 1) or (q == SPT - 1)
        SA = [[sum(mm(S[k][a], A[k][b]) for k in R) for b in R] for a in R]
        SB = [sum(mm(S[k][a], Bp[k]) for k in R) for a in R]
        F = [sum(mm(Bp[k], SA[k][b]) for k in R) for b in R]
        Re = sum(mm(Bp[k], SB[k]) for k in R) + Rp
        R0 = small_inverse(Re, nup)
        K = [mm(R0, F[b]) for b in R]
        Acl = [[mm(nBt[a], K[b], A[a][b]) for b in R] for a in R]
        if j > 0:
            Zm = [[sum(mm(S[k][a], Acl[k][b]) for k in R) for b in R] for a in R]
            S = [[Q[a][b] + sum(mm(A[k][a], Zm[k][b]) for k in R) for b in R] for a in R]
        SH = Z4(); Iq = Z4()
        for u in range(nup): SH[u, off + u] = 1.0; Iq[off + u, off + u] = 1.0
        R0h = 0.5 * R0
        Ktp = [mm(K[a], SH) for a in R]
        R1h = R0h if off == 0 else mm(R0h, SH)
        Rqq = R0h if off == 0 else mm(SH, R1h)
        nBRt = [mm(nBt[a], R1h) for a in R]
        live = lambda I: min((I + 1) * SPT, N) - 1 > j
        for I in range(Ij, NT):
            cT = (0 if first else TtA[I]) + (Iq if I == Ij else 0)
            cD = (0 if first else TDA[I]) + (Rqq if I == Ij else 0)
            if live(I):
                TtA[I] = sum(mm(nBpl[q][k], rho[I][k]) for k in R) + cT
                TDA[I] = sum(mm(nBRt[k], rho[I][k]) for k in R) + cD
            else:
                TtA[I], TDA[I] = cT + Z4(), cD + Z4()
        if q == 0:
            for I in range(Ij, NT):
                for J in range(Ij, I + 1): W[I, J] = mm(TDA[I], TtA[J], W[I, J])
        for I in range(Ij, NT):
            if live(I): rho[I] = [sum(mm(Acl[k][a], rho[I][k]) for k in R) + (Ktp[a] if I == Ij else 0) for a in R]
            else: rho[I] = [Ktp[a].copy() for a in R]
    n = N * nu
    rmap = [(rp // nup) * nu + rp % nup if rp % nup < nu and rp < npad else -1 for rp in range(4 * NT)]
    Wd = np.zeros((n, n)); G = np.zeros((n, nx))
    for (I, J), t in W.items():
        for r in range(4):
            for c in range(4):
                ri, ci = rmap[4 * I + r], rmap[4 * J + c]
                if ri >= 0 and ci >= 0 and ci <= ri: Wd[ri, ci] = t[r, c]; Wd[ci, ri] = t[r, c]
    for I in range(NT):
        for a in R:
            for r in range(4):
                for c in range(4):
                    ri, st = rmap[4 * I + c], 4 * a + r
                    if ri >= 0 and st < nx: G[ri, st] = -rho[I][a][r, c]
    # P, Toeplitz form
    Lt = sq(PT_)
    ap = {}
    for k in range(N - 1, -1, -1):
        ap[k] = [sum(mm(Lt[k2][a], Bpl[k % SPT][k2]) for k2 in R) for a in R]
        if k > 0:
            LA = [[sum(mm(Lt[k2][a], A[k2][b]) for k2 in R) for b in R] for a in R]
            Lt = [[Q[a][b] + sum(mm(A[k2][a], LA[k2][b]) for k2 in R) for b in R] for a in R]
    Rd = Z4()
    for qq in range(SPT): Rd[qq * nup:qq * nup + nup, qq * nup:qq * nup + nup] = Rp[:nup, :nup]
    X = [t.copy() for t in Bpl[0]]
    Pd = np.zeros((n, n))
    for D in range(NT):
        acc = {I: (Rd.copy() if D == 0 else Z4()) for I in range(D, NT)}
        for pp in range(SPT):
            d = D * SPT + pp
            if d > 0:
                X = [sum(mm(At[k][a], X[k]) for k in R) + (Bpl[d][a] if d <= SPT - 1 else 0) for a in R]
            for I in range(D, NT):
                k = I * SPT + pp
                if k < N: acc[I] = acc[I] + sum(mm(ap[k][k2], X[k2]) for k2 in R)
        for I in range(D, NT):
            for r in range(4):
                for c in range(4):
                    ri, ci = rmap[4 * I + r], rmap[4 * (I - D) + c]
                    if ri >= 0 and ci >= 0 and ci <= ri: Pd[ri, ci] = Pd[ci, ri] = 2.0 * acc[I][r, c]
    return Wd, G, Pd

def check(nx, nu, N, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A = q @ np.diag(rng.uniform(0.6, 1.1, nx)); B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    M = rng.standard_normal((nx, nx)); Q = 2 * np.eye(nx) + 0.1 * M @ M.T
    M = rng.standard_normal((nu, nu)); R = np.eye(nu) + 0.1 * M @ M.T
    M = rng.standard_normal((nx, nx)); PT = 3 * np.eye(nx) + 0.2 * M @ M.T
    W, G, P = setup(A, B, Q, R, PT, N)
    H, F = synth.condense_np(A, B, Q, R, PT, N)
    Wr = np.linalg.inv(2 * H); Gr = -np.linalg.solve(H, F)
    e = [np.max(np.abs(W - Wr)) / np.max(np.abs(Wr)), np.max(np.abs(G - Gr)) / np.max(np.abs(Gr)), np.max(np.abs(P - 2 * H)) / np.max(np.abs(H))]
    print(f"nx={nx} nu={nu} N={N}: relerr W {e[0]:.1e} G {e[1]:.1e} P {e[2]:.1e}")
    assert max(e) < 1e-9

if __name__ == "__main__":
    for (nx, nu, N) in [(4, 2, 10), (2, 1, 7), (5, 3, 4), (8, 4, 8), (6, 3, 15), (5, 1, 33), (7, 2, 9), (3, 3, 5), (8, 4, 12), (4, 4, 6), (1, 1, 1), (8, 1, 48), (6, 2, 24)]:
        check(nx, nu, N, 11 + nx + nu + N)
