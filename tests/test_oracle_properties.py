"""CPU: independent checks of the oracle's exact box-QP solver and condensing."""
import numpy as np
import pytest
from scipy.optimize import lsq_linear

from oracle import oracle as orc
from lq_mpc_amd import synth


def _kkt_violation(H, g, lb, ub, u):
    grad = 2 * (H @ u + g)
    tol = 1e-9
    free = (u > lb + tol) & (u < ub - tol)
    v = np.max(np.abs(grad[free])) if free.any() else 0.0
    at_lb, at_ub = u <= lb + tol, u >= ub - tol
    if at_lb.any():
        v = max(v, np.max(np.maximum(-grad[at_lb], 0)))
    if at_ub.any():
        v = max(v, np.max(np.maximum(grad[at_ub], 0)))
    return v


@pytest.mark.parametrize("n,seed", [(1, 0), (5, 1), (10, 2), (20, 3), (40, 4), (120, 5)])
def test_boxqp_vs_bvls_and_kkt(n, seed):
    rng = np.random.default_rng(seed)
    for trial in range(6):
        M = rng.standard_normal((n + 3, n))
        H = M.T @ M + 0.1 * np.eye(n)
        g = rng.standard_normal(n) * (10.0 ** rng.integers(-2, 2))
        lb, ub = -rng.uniform(0.05, 1, n), rng.uniform(0.05, 1, n)
        u, it = orc.boxqp(H, g, lb, ub)
        assert np.all(u >= lb) and np.all(u <= ub)
        assert _kkt_violation(H, g, lb, ub, u) < 1e-8 * max(1, np.abs(g).max())
        # min u'Hu + 2g'u == min |L'u + L^-1 g|^2
        L = np.linalg.cholesky(H)
        ref = lsq_linear(L.T, -np.linalg.solve(L, g), bounds=(lb, ub), method="bvls", tol=1e-14).x
        np.testing.assert_allclose(u, ref, atol=1e-9)


def test_boxqp_edge_cases():
    H = np.array([[2.0, 0.5], [0.5, 1.0]])
    lb, ub = np.array([-0.1, -0.1]), np.array([0.1, 0.1])
    u, _ = orc.boxqp(H, np.zeros(2), lb, ub)                    # g = 0 -> u = 0
    assert np.all(u == 0)
    u, _ = orc.boxqp(H, np.array([100.0, 100.0]), lb, ub)       # everything at the lower bound
    np.testing.assert_array_equal(u, lb)
    u, _ = orc.boxqp(H, np.array([-100.0, 100.0]), lb, ub)
    np.testing.assert_array_equal(u, [0.1, -0.1])
    u, _ = orc.boxqp(H, np.array([1e-6, -2e-6]), lb, ub)        # strictly interior: u = -H^-1 g
    np.testing.assert_allclose(u, -np.linalg.solve(H, [1e-6, -2e-6]), rtol=1e-13)


def test_condense_against_reference_Gamma_formula():
    """H from the oracle equals Gamma' Qbar Gamma + Rbar built the way utils.py:145-174, 316-319 do."""
    rng = np.random.default_rng(7)
    nx, nu, N = 3, 2, 5
    A, B = rng.standard_normal((nx, nx)) * 0.5, rng.standard_normal((nx, nu))
    Q, R, P = np.diag([2.0, 1.0, 3.0]), np.diag([1.0, 0.5]), np.diag([4.0, 4.0, 1.0])
    H, F = orc.condense(A, B, Q, R, P, N)
    G = np.zeros((N * nx, N * nu))
    for r in range(N):
        for c in range(r + 1):
            G[r * nx:(r + 1) * nx, c * nu:(c + 1) * nu] = np.linalg.matrix_power(A, r - c) @ B
    Phi = np.vstack([np.linalg.matrix_power(A, k) for k in range(1, N + 1)])
    Qb = np.kron(np.eye(N), Q)
    Qb[-nx:, -nx:] = P
    np.testing.assert_allclose(H, G.T @ Qb @ G + np.kron(np.eye(N), R), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(F, G.T @ Qb @ Phi, rtol=1e-12, atol=1e-13)


def test_solve_with_references_matches_bruteforce_cost():
    rng = np.random.default_rng(11)
    nx, nu, N = 3, 2, 4
    A, B = rng.standard_normal((nx, nx)) * 0.6, rng.standard_normal((nx, nu))
    Q, R, P = np.diag([2.0, 1.0, 3.0]), np.diag([1.0, 0.5]), np.diag([4.0, 4.0, 1.0])
    lb, ub = np.array([-0.3, -0.2]), np.array([0.1, 0.4])
    x0 = rng.standard_normal(nx)
    xr, ur = rng.standard_normal((nx, N)) * 0.3, rng.standard_normal((nu, N)) * 0.1
    out = orc.solve(N, A, B, Q, R, P, lb, ub, x0, xr, ur)

    def cost(U):   # utils_class.py:59-75 literally
        x, c = x0.copy(), 0.0
        for i in range(N):
            x = A @ x + B @ U[:, i]
            W = Q if i < N - 1 else P
            c += (x - xr[:, i]) @ W @ (x - xr[:, i]) + (U[:, i] - ur[:, i]) @ R @ (U[:, i] - ur[:, i])
        return c

    assert abs(cost(out["U"]) + x0 @ Q @ x0 - out["V_N"]) < 1e-11
    for _ in range(200):   # no feasible perturbation improves the cost
        U2 = np.clip(out["U"] + 1e-3 * rng.standard_normal((nu, N)), lb[:, None], ub[:, None])
        assert cost(U2) >= cost(out["U"]) - 1e-12


def test_simulate_equals_stepwise_solve():
    b = synth.make_batch(3, Bsz=4)
    N, T = b["N"], 6
    for i in range(4):
        A, B, x = b["A"][:, :, i], b["B"][:, :, i], b["x0"][:, i].copy()
        sim = orc.simulate(T, N, A, B, b["Q"], b["R"], b["P"], b["lb"], b["ub"], x, b["A_true"], b["B_true"])
        J = x @ b["Q"] @ x
        for t in range(T):
            u = orc.solve(N, A, B, b["Q"], b["R"], b["P"], b["lb"], b["ub"], x)["u_0"]
            x = b["A_true"] @ x + b["B_true"] @ u
            J += x @ b["Q"] @ x + u @ b["R"] @ u
            np.testing.assert_allclose(sim["U"][:, t], u, atol=1e-13)
        assert abs(J - sim["J_T"]) < 1e-12 * max(1, abs(J))


def test_synth_is_seeded_and_shaped():
    a, b = synth.make_batch(3, Bsz=128), synth.make_batch(3, Bsz=128)
    for k in ("A", "B", "x0"):
        np.testing.assert_array_equal(a[k], b[k])
    assert a["A"].shape == (4, 4, 128) and a["B"].shape == (4, 2, 128) and a["x0"].shape == (4, 128)
    assert np.all(np.linalg.norm((a["A"] - a["A_true"][:, :, None]).reshape(16, -1), axis=0) <= 1e-2 + 1e-15)
