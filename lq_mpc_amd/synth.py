"""Seeded synthetic batches of perturbed LTI systems.

Common to every config: Q = 2 I, R = I, P = Q, zero references, box |u_k| <= 0.1 -- the
reference's own values (/root/reference/working_example_multiple.py:13-25) -- fp64, RNG
numpy.random.default_rng(20250404 + config_id).  Arrays are returned in the library's
instance-minor (SoA) layout: A (nx, nx, Bsz), B (nx, nu, Bsz), x0 (nx, Bsz); that is the
layout of the reference's error_{A,B}_f.npy files (entry-major, instance-minor).
"""
import os

import numpy as np

CONFIGS = {
    # id: (nx, nu, N, Bsz, T, rho_lo, rho_hi)
    1: dict(nx=2, nu=1, N=5, Bsz=1, T=50),
    2: dict(nx=2, nu=1, N=10, Bsz=4096, T=30),
    3: dict(nx=4, nu=2, N=10, Bsz=65536, T=30, rho=(0.7, 1.1)),
    4: dict(nx=4, nu=2, N=20, Bsz=262144, T=30, rho=(0.7, 1.1)),
    5: dict(nx=8, nu=4, N=30, Bsz=32768, T=30, rho=(0.6, 1.05)),
}

A_REF = np.array([[1.0, 0.7], [0.12, 0.4]])   # working_example_multiple.py:13
B_REF = np.array([[1.0], [1.2]])              # working_example_multiple.py:14
U_MAX = 0.1                                   # F_u = [10 I; -10 I], working_example_multiple.py:25
X_START_GLOBAL = np.array([0.15916231240837822, 0.15916231240837819])  # x0_vec[:, 1]


def _dlqr_gain(A, B, Q, R):
    from scipy.linalg import solve_discrete_are
    Pinf = solve_discrete_are(A, B, Q, R)
    return np.linalg.solve(R + B.T @ Pinf @ B, B.T @ Pinf @ A)


def _frobenius_ball(rng, shape, Bsz, delta):
    """One sample per instance, uniform in the Frobenius ball of radius delta[b]."""
    d = int(np.prod(shape))
    g = rng.standard_normal((d, Bsz))
    g /= np.linalg.norm(g, axis=0, keepdims=True)
    r = delta * rng.random(Bsz) ** (1.0 / d)
    return (g * r).reshape(*shape, Bsz)


def base_system(config_id):
    """The unperturbed (A, B) of a config: the plant used in rollout mode."""
    c = CONFIGS[config_id]
    nx, nu = c["nx"], c["nu"]
    if nx == 2:
        return A_REF.copy(), B_REF.copy()
    rng = np.random.default_rng(20250404 + config_id)
    q, r = np.linalg.qr(rng.standard_normal((nx, nx)))
    q = q * np.sign(np.diag(r))
    rho = rng.uniform(c["rho"][0], c["rho"][1], nx)
    A = q @ np.diag(rho)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    return A, B


HARD_S = (6.0, 24.0)   # "hard" mix: the unconstrained LQR input at step 0 is s*u_max with s ~ U[6, 24] (default mix: U[0.5, 4])


def make_batch(config_id, Bsz=None, fixture_dir=None, mix="default"):
    """Returns a dict with the batch in SoA layout plus the shared cost/box/plant data.

    mix = "hard": same models, initial states 6-24x outside the region where the box is inactive, so that most MPC steps of a
    T = 30 rollout are constrained QPs (the default mix of SURVEY 8(d) leaves ~85 % of the steps to the presolve)."""
    c = CONFIGS[config_id]
    nx, nu, N = c["nx"], c["nu"], c["N"]
    Bsz = int(Bsz or c["Bsz"])
    Q, R = 2.0 * np.eye(nx), np.eye(nu)
    A0, B0 = base_system(config_id)
    rng = np.random.default_rng(20250404 + config_id + 1000)
    out = dict(config_id=config_id, nx=nx, nu=nu, N=N, T=c["T"], Bsz=Bsz, Q=Q, R=R, P=Q.copy(),
               lb=np.full(nu, -U_MAX), ub=np.full(nu, U_MAX), A_true=A0, B_true=B0)
    if config_id == 1:
        out["A"] = np.repeat(A0[:, :, None], Bsz, 2)
        out["B"] = np.repeat(B0[:, :, None], Bsz, 2)
        out["x0"] = np.repeat(X_START_GLOBAL[:, None], Bsz, 1)
        return out
    delta = rng.uniform(1e-3, 1e-2, Bsz)
    dA = _frobenius_ball(rng, (nx, nx), Bsz, delta)
    dB = _frobenius_ball(rng, (nx, nu), Bsz, delta)
    if config_id == 2 and fixture_dir is not None:
        # first 1000 instances are the reference's shipped perturbations (utils_class.py:749-750)
        eA = np.load(os.path.join(fixture_dir, "error_A_f.npy")).reshape(nx, nx, -1)
        eB = np.load(os.path.join(fixture_dir, "error_B_f.npy")).reshape(nx, nu, -1)
        m = min(Bsz, eA.shape[2])
        dA[:, :, :m] = eA[:, :, :m]
        dB[:, :, :m] = eB[:, :, :m]
    out["A"] = np.ascontiguousarray(A0[:, :, None] + dA)
    out["B"] = np.ascontiguousarray(B0[:, :, None] + dB)
    # x0: uniform direction, scaled so the unconstrained LQR input at step 0 is s*u_max, s ~ U[0.5, 4]
    K = _dlqr_gain(A0, B0, Q, R)
    d = rng.standard_normal((nx, Bsz))
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    s = rng.uniform(0.5, 4.0, Bsz) if mix == "default" else rng.uniform(HARD_S[0], HARD_S[1], Bsz)
    ku = np.max(np.abs(K @ d), axis=0)
    out["x0"] = np.ascontiguousarray(d * (s * U_MAX / ku))
    out["mix"] = mix
    return out


def condense_np(A, B, Q, R, P, N):
    """H = G'QbarG + Rbar and F = G'Qbar Phi of one model in numpy; cost = U'HU + 2(F x0)'U + c."""
    nx, nu = B.shape
    pw = [np.eye(nx)]
    for _ in range(N):
        pw.append(A @ pw[-1])
    G = np.zeros((N * nx, N * nu))
    for r in range(N):
        for c in range(r + 1):
            G[r * nx:(r + 1) * nx, c * nu:(c + 1) * nu] = pw[r - c] @ B
    Phi = np.vstack(pw[1:])
    Qb = np.kron(np.eye(N), Q)
    Qb[-nx:, -nx:] = P
    return G.T @ Qb @ G + np.kron(np.eye(N), R), G.T @ Qb @ Phi


def constrained_share(batch, X, idx):
    """Share of the MPC steps along the trajectories X (nx, T+1, len(idx)) of instances `idx` whose QP has an unconstrained
    minimiser outside the box, i.e. the steps the presolve cannot finish (zero references)."""
    lb, ub = np.tile(batch["lb"], batch["N"])[:, None], np.tile(batch["ub"], batch["N"])[:, None]
    cons = tot = 0
    for j, i in enumerate(idx):
        H, F = condense_np(batch["A"][:, :, i], batch["B"][:, :, i], batch["Q"], batch["R"], batch["P"], batch["N"])
        v = -np.linalg.solve(H, F @ X[:, :-1, j])
        cons += int(((v < lb) | (v > ub)).any(axis=0).sum())
        tot += X.shape[1] - 1
    return cons / max(tot, 1)
