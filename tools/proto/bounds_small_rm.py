"""Dev prototype (numpy, CPU): the small-matrix half of the bound coefficients (dlqr by doubling, spectral radius and spectral norms
by repeated squaring) written with D = A'B + C products only, as lqmpc_bounds_chip.h does on the matrix core."""
import numpy as np
from scipy.linalg import solve_discrete_are

def mm(a, b, c=None):
    d = a.T @ b
    return d if c is None else d + c

def log_rho_by_squaring(Y, Yt, J=56):
    """log of the spectral radius of Y (Yt = Y'): sum_j 2^-j log |Y_j|_F, Y_{j+1} = (Y_j / |Y_j|)^2"""
    lg, w = 0.0, 1.0
    for _ in range(J):
        s = np.sqrt(np.sum(Y * Y))
        if not s > 0: return -np.inf
        lg += w * np.log(s); w *= 0.5
        Y, Yt = Y / s, Yt / s
        Y, Yt = mm(Yt, Y), mm(Y, Yt)          # Y Y and (Y Y)' = Y' Y'
    return lg

def dlqr_doubling(A, B, Q, R):
    nx = A.shape[0]
    Ak, Akt = A.copy(), A.T.copy()
    G = B @ np.linalg.inv(R) @ B.T; H = Q.copy()
    I = np.eye(nx)
    for it in range(64):
        Hi = np.linalg.inv(H); S = np.linalg.inv(Hi + G)
        W1 = mm(Hi, S)                      # Hi S
        Vt = mm(W1, Akt)                    # S Hi Ak' = V'
        GVt = mm(G, Vt)                     # G V' = (V G)'
        Gn = mm(GVt, Akt, G)                # V G Ak' + G
        SA = mm(S, Ak)                      # S Ak
        Hn = mm(Ak, SA, H)                  # Ak' S Ak + H
        Akn, Aktn = mm(Vt, Ak), mm(Ak, Vt)  # V Ak, Ak' V'
        dn = np.sum((Hn - H) ** 2); hn = np.sum(Hn * Hn)
        Ak, Akt, G, H = Akn, Aktn, 0.5 * (Gn + Gn.T), 0.5 * (Hn + Hn.T)
        if dn <= 1e-34 * hn: break
    P = H
    SA, SB = mm(P, A), mm(P, B)
    K = np.linalg.solve(R + mm(B, SB), mm(B, SA))
    return K, P, it

rng = np.random.default_rng(3)
for nx, nu in [(2, 1), (4, 2), (5, 3), (8, 4)]:
    A = rng.standard_normal((nx, nx)); A *= 1.05 / np.max(np.abs(np.linalg.eigvals(A)))
    B = rng.standard_normal((nx, nu))
    M = rng.standard_normal((nx, nx)); Q = M @ M.T / nx + np.eye(nx)
    M = rng.standard_normal((nu, nu)); R = M @ M.T / nu + np.eye(nu)
    K, P, it = dlqr_doubling(A, B, Q, R)
    Pr = solve_discrete_are(A, B, Q, R); Kr = np.linalg.solve(R + B.T @ Pr @ B, B.T @ Pr @ A)
    Acl = A - B @ K
    rho = np.exp(log_rho_by_squaring(Acl, Acl.T.copy()))
    nA = np.exp(0.5 * log_rho_by_squaring(mm(A, A), mm(A, A)))
    print(nx, nu, "iters", it, "K err", np.max(np.abs(K - Kr)) / np.max(np.abs(Kr)), "rho err", abs(rho - np.max(np.abs(np.linalg.eigvals(Acl)))),
          "|A|2 err", abs(nA - np.linalg.norm(A, 2)))
