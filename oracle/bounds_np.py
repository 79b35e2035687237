"""TEST INFRASTRUCTURE (host numpy oracle of lqmpc_bounds_batch; only tests/ may import it).

Scalar coefficients of the reference's performance bound: restates, one system at a time, what
LQ_RDP_Calculator.energy_decreasing / energy_bound (/root/reference/utils_class.py:308-373) compute
through /root/reference/utils.py:
    g functions                    utils.py:78-117      stage(g_x, g_u)
    bar-g sums                     utils.py:186-223
    theta                          utils.py:226-264
    E_psi, E_u, E_psi_u            utils.py:267-334
    exponential-stability numbers  utils.py:343-380
    omega_N1, omega_N0.5, eta      utils.py:469-523     (geo_M 393-409)
    h                              utils.py:526-538
    L_V, N_0                       utils.py:567-584
    bar_u, bar_d_u                 utils.py:592-650     (two Gurobi toy QPs there; closed form for a box here)
The product computes the same numbers on the GPU (lq_mpc_amd/csrc/lqmpc_bounds.hip); this file is what that kernel is checked
against, and it is itself pinned by the reference's own npz (tests/test_bounds.py).  Two quirks of the reference are kept on purpose because the
golden data contain them: the constant lambda_K = 1.21 and the "+0.4" in rho_K (utils.py:358, 364).
One is NOT kept: the reference forms the closed loop as `A + B * K` (elementwise, utils.py:356), which
equals A + B @ K only for n_u = 1; this module uses the matrix product (identical on the golden data).
"""
# parity: pinned (alpha/beta/xi/bound tables of data_lq_mpc_multipleSys.npz, tests/test_bounds.py)
import math

import numpy as np


def _eig_info(M):
    ev = np.linalg.eigvals(M).real
    return float(ev.max()), float(ev.min())


def box_bar_u(lb, ub):
    """max |u|^2 and max |u1 - u2|^2 over the box (utils.py:592-650 solve these with Gurobi)."""
    lb, ub = np.asarray(lb, float), np.asarray(ub, float)
    return float(np.sum(np.maximum(lb ** 2, ub ** 2))), float(np.sum((ub - lb) ** 2))


def g_x(power, i, e_A, f_A):
    return ((e_A + f_A) ** i - f_A ** i) ** power                                   # utils.py:78-95


def g_u(power, i, e_A, f_A, e_B, f_B):
    return ((e_B + f_B) * g_x(1, i, e_A, f_A) + e_B * f_A ** i) ** power             # utils.py:98-117


def gamma_phi(N, A, B):
    """Gamma with the reference's extra zero block row ((N+1) n_x rows) and Phi = [I; A; ...; A^N] (utils.py:126-174)."""
    nx, nu = B.shape
    G = np.zeros(((N + 1) * nx, N * nu))
    pw = [np.linalg.matrix_power(A, k) for k in range(N + 1)]
    for r in range(1, N + 1):
        for c in range(r):
            G[r * nx:(r + 1) * nx, c * nu:(c + 1) * nu] = pw[r - 1 - c] @ B
    return G, np.vstack(pw)


def stability_numbers(A, B, Q, R, K):
    """C*_K, gamma, rho_gamma of utils.py:343-380 (K is the gain with u = K x, i.e. the caller passes -K_lqr)."""
    qmax, qmin = _eig_info(Q)
    rmax, _ = _eig_info(R)
    rho_K = (np.max(np.abs(np.linalg.eigvals(A + B @ K))) + 0.4) ** 2
    C_star = (1 + rmax * np.linalg.norm(K, 2) ** 2 / qmin) * max(1.0, qmax / qmin * 1.21)
    gamma = C_star / (1 - rho_K)
    return {"C_K": C_star, "rho_K": rho_K, "gamma": gamma, "rho_gamma": (gamma - 1) / gamma}


def local_radius(F_u, K, Q):
    M = np.asarray(F_u) @ np.asarray(K)
    invQ = np.linalg.inv(Q)
    return 1.0 / max(float(M[i] @ invQ @ M[i]) for i in range(M.shape[0]))


def energy_decreasing(N, A, B, Q, R, F_u, e_A, e_B, K, M_V):
    """xi and eta of utils_class.py:344-373 for the model (A, B) with gain K (u = K x) and energy bar M_V."""
    qmax, qmin = _eig_info(Q)
    _, rmin = _eig_info(R)
    eps = local_radius(F_u, K, Q)
    st = stability_numbers(A, B, Q, R, K)
    L_V = max(st["gamma"], M_V / eps)                                                # utils.py:575
    N_0 = math.ceil(max(0.0, M_V / eps - st["gamma"]))                               # utils.py:576
    nA = np.linalg.norm(A, 2)
    G_A = float(N - 1) if nA == 1 else (1 - nA ** (2 * (N - 1))) / (1 - nA ** 2)     # utils.py:393-409
    term = 1 + nA ** 2 * qmax / qmin                                                 # utils.py:503
    omega_1 = qmax * (term * nA ** (2 * N - 2) + G_A)                                # utils.py:510
    decay = qmax * nA ** (2 * N - 2) * st["gamma"] * st["rho_gamma"] ** (N - N_0)
    omega_05 = math.sqrt(qmax * (L_V - 1) * G_A) + 0.5 * term * math.sqrt(decay)     # utils.py:514
    eta = (term - 1) * st["gamma"] * st["rho_gamma"] ** (N - N_0)                    # utils.py:517
    h = e_A ** 2 / qmin + e_B ** 2 / rmin                                            # utils.py:538
    return {"xi": h * omega_1 + 2 * math.sqrt(h) * omega_05, "eta": eta, "L_V": L_V, "N_0": N_0}


def energy_bound(N, A, B, Q, R, lb, ub, e_A, e_B, x, p):
    """alpha and beta of utils_class.py:308-342."""
    bar_u, bar_d_u = box_bar_u(lb, ub)
    qmax, _ = _eig_info(Q)
    rmax, _ = _eig_info(R)
    f_A, f_B = np.linalg.norm(A, 2), np.linalg.norm(B, 2)
    nx2 = float(np.linalg.norm(x, 2)) ** 2
    s_in = s_out = 0.0
    for i in range(N + 1):                                                           # utils.py:296-302
        s_out += (s_in + g_x(2, i, e_A, f_A)) * (nx2 + i * bar_u)
        s_in += g_u(2, i, e_A, f_A, e_B, f_B)
    E_psi = qmax * s_out
    G, Phi = gamma_phi(N, A, B)
    nG, nPhi = np.linalg.norm(G, 2), np.linalg.norm(Phi, 2)
    bar_gx = sum(g_x(1, i + 1, e_A, f_A) for i in range(N))                          # utils.py:186-201
    bar_gu = s = 0.0
    for i in range(N):                                                               # utils.py:204-223
        s += g_u(1, i, e_A, f_A, e_B, f_B)
        bar_gu += s
    theta_u = qmax * (2 * nG * bar_gu + bar_gu ** 2)                                 # utils.py:253-255
    theta_xu = qmax * (nG * bar_gx + nPhi * bar_gu + bar_gx * bar_gu)                # utils.py:258-262
    bar_theta = math.sqrt(N * bar_u) * theta_u + math.sqrt(nx2) * theta_xu           # utils.py:313
    hatH = np.kron(R, np.eye(N)) + G.T @ np.kron(Q, np.eye(N + 1)) @ G               # utils.py:316-319 (ordering as there)
    min_H = float(np.min(np.linalg.eigvals(hatH).real))
    E_u = rmax * min(math.sqrt(N * bar_d_u), bar_theta / min_H) ** 2                 # utils.py:325
    E_psi_u = qmax / rmax * (nG + bar_gu) ** 2 * E_u                                 # utils.py:331
    p = np.asarray(p, float)
    q = 1.0 / p
    sp, su, spu = math.sqrt(E_psi), math.sqrt(E_u), math.sqrt(E_psi_u)
    alpha = max(p[0] * sp + p[2] * spu + p[0] * sp * p[2] * spu, p[1] * su)          # utils_class.py:332-335
    beta = (1 + p[0] * sp) * (q[2] * spu + E_psi_u) + q[1] * su + E_u + q[0] * sp + E_psi   # utils_class.py:338-340
    return {"alpha": alpha, "beta": beta}
