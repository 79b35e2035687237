"""Dev prototype (numpy, CPU): the set-up of the 16-lane-row kernels restated as 4x4-tile products of the form D = A'B + C --
what one v_mfma_f64_4x4x4_4b_f64 computes per 16-lane group when every operand is a 4x4 matrix held one element per lane
(lane t <-> entry [t/4][t%4]).  Checks W = P^-1, G = -P^-1 Fq and P against the dense condensing of lq_mpc_amd.synth.condense_np.

    backward Riccati sweep (stage j = N-1 .. 0):  Sg = Q + S_j (cost-to-go incl. stage cost), Re, K, Acl
    rho_k(j) = K_k Acl_{k-1} .. Acl_{j+1}  (k > j):  T_{k,j} = -rho_k(j) B,  W = 1/2 T D^-1 T' accumulated as rank-NU updates per stage,
    G rows = -rho_k(-1);  P = 2 (Gamma'Qbar Gamma + Rbar) from the Lyapunov/Toeplitz form, tile by tile.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from lq_mpc_amd import synth

NMM = [0]
def mm(a, b, c=None):
    NMM[0] += 1
    d = a.T @ b
    return d if c is None else d + c

def pad(M, r0=0, c0=0):
    out = np.zeros((4, 4)); out[r0:r0 + M.shape[0], c0:c0 + M.shape[1]] = M; return out

def setup_tiles(A_, B_, Q_, R_, PT_, N):
    nx, nu = B_.shape
    assert nx <= 4 and 4 % nu == 0
    n = N * nu; NT = (n + 3) // 4; SPT = 4 // nu
    A = pad(A_); At = pad(A_.T); Bp = pad(B_); Bt = pad(B_.T); Q = pad(Q_); PT = pad(PT_); Rp = pad(R_)
    S = PT.copy()
    rho = [np.zeros((4, 4)) for _ in range(NT)]
    W = {(I, J): np.zeros((4, 4)) for I in range(NT) for J in range(I + 1)}
    spd = True
    for j in range(N - 1, -1, -1):
        Ij, off = (j * nu) // 4, (j * nu) % 4
        SA = mm(S, A); SB = mm(S, Bp)
        F = mm(Bp, SA); Re = mm(Bp, SB, Rp)
        Ri = np.linalg.inv(Re[:nu, :nu]); spd = spd and np.all(np.linalg.eigvalsh(Re[:nu, :nu]) > 0)
        R0 = pad(Ri); R1 = pad(Ri, 0, off)
        K = mm(R0, F); Ktp = mm(F, R1)
        Acl = mm(-Bt, K, A)
        Z = mm(S, Acl); S = mm(A, Z, Q)
        BRt = mm(Bt, 0.5 * R0)
        SH = np.zeros((4, 4))
        for u in range(nu): SH[u, off + u] = 1.0
        Tt, TDt = {}, {}
        for I in range(Ij, NT):
            Tt[I] = -mm(Bp, rho[I]); TDt[I] = -mm(BRt, rho[I])
        Tt[Ij] = Tt[Ij] + SH; TDt[Ij] = TDt[Ij] + 0.5 * R1
        for I in range(Ij, NT):
            for J in range(Ij, I + 1):
                W[I, J] = mm(TDt[I], Tt[J], W[I, J])
        for I in range(Ij, NT): rho[I] = mm(Acl, rho[I])
        rho[Ij] = rho[Ij] + Ktp
    Wd = np.zeros((4 * NT, 4 * NT)); G = np.zeros((4 * NT, 4))
    for (I, J), t in W.items(): Wd[4 * I:4 * I + 4, 4 * J:4 * J + 4] = t
    Wd = np.tril(Wd) + np.tril(Wd, -1).T
    for I in range(NT): G[4 * I:4 * I + 4, :] = -rho[I].T
    # P = 2 (H + Rbar), Toeplitz form: block (bi, bj) = B' Lt_{bi+1} A^(bi-bj) B
    Lt = PT.copy()
    ap = {}                                   # a-operands: Lt_{bi+1} B placed at the columns of stage bi inside its tile
    for k in range(N - 1, -1, -1):
        ap[k] = mm(Lt, pad(B_, 0, (k * nu) % 4))          # Lt symmetric: Lt' = Lt
        if k > 0: Lt = mm(A, mm(Lt, A), Q)
    X = {0: Bp.copy()}
    for d in range(1, N + SPT):
        X[d] = mm(At, X[d - 1])
        if d <= SPT - 1: X[d] = X[d] + pad(B_, 0, d * nu)
    Rd = np.zeros((4, 4))
    for p in range(SPT): Rd[p * nu:(p + 1) * nu, p * nu:(p + 1) * nu] = R_
    Pd = np.zeros((4 * NT, 4 * NT))
    for I in range(NT):
        for J in range(I + 1):
            t = Rd.copy() if I == J else np.zeros((4, 4))
            for p in range(SPT):
                k = I * SPT + p
                if k < N: t = mm(ap[k], X[(I - J) * SPT + p], t)
            Pd[4 * I:4 * I + 4, 4 * J:4 * J + 4] = 2.0 * t
    Pd = np.tril(Pd) + np.tril(Pd, -1).T
    return Wd[:n, :n], G[:n, :nx], Pd[:n, :n], spd

def check(nx, nu, N, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A = q @ np.diag(rng.uniform(0.6, 1.1, nx)); B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    M = rng.standard_normal((nx, nx)); Q = 2 * np.eye(nx) + 0.1 * M @ M.T
    M = rng.standard_normal((nu, nu)); R = np.eye(nu) + 0.1 * M @ M.T
    M = rng.standard_normal((nx, nx)); PT = 3 * np.eye(nx) + 0.2 * M @ M.T
    NMM[0] = 0
    W, G, P, spd = setup_tiles(A, B, Q, R, PT, N)
    H, F = synth.condense_np(A, B, Q, R, PT, N)
    Wr = np.linalg.inv(2 * H); Gr = -np.linalg.solve(H, F)
    e = [np.max(np.abs(W - Wr)) / np.max(np.abs(Wr)), np.max(np.abs(G - Gr)) / np.max(np.abs(Gr)), np.max(np.abs(P - 2 * H)) / np.max(np.abs(H))]
    print(f"nx={nx} nu={nu} N={N}: relerr W {e[0]:.1e} G {e[1]:.1e} P {e[2]:.1e}  cond(H) {np.linalg.cond(H):.1e}  tile products {NMM[0]} spd {spd}")
    assert max(e) < 1e-9

if __name__ == "__main__":
    for (nx, nu, N) in [(4, 2, 10), (2, 1, 10), (2, 1, 5), (2, 1, 7), (2, 1, 30), (4, 2, 20), (3, 2, 6), (1, 1, 1), (4, 4, 6), (3, 1, 10), (4, 2, 12), (2, 2, 3), (4, 1, 9)]:
        check(nx, nu, N, 7 + nx + nu + N)
