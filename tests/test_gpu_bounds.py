"""GPU (-m gpu): lqmpc_bounds_batch -- dlqr + energy_decreasing + energy_bound per system on the device
-- through the C ABI, against the host oracle (oracle/bounds_np.py, scipy's DARE) and the reference's npz.

Tolerances: the numbers are smooth functions of eigenvalues / singular values computed by different methods on the two sides
(doubling + Jacobi sweeps + repeated squaring on the GPU; LAPACK on the host): 1e-9 relative, 1e-8 against the npz as in
tests/test_bounds.py."""
import os

import numpy as np
import pytest

from oracle import bounds_np
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

A0 = np.array([[1.0, 0.7], [0.12, 0.4]])
B0 = np.array([[1.0], [1.2]])
Q2 = 2.0 * np.eye(2)
R1 = np.eye(1)
F_U = np.vstack((10 * np.eye(1), -10 * np.eye(1)))
P3 = np.array([0.1, 1, 0.6])


def rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


CHIP = "lqmpc_bounds_small_kernel + lqmpc_bounds_big_kernel"     # registers + LDS (prebuilt or compiled at run time); the default
WORKSPACE = "lqmpc_bounds_kernel"                                # one instance per lane, matrices in an HBM workspace (kernel = generic)


@pytest.fixture(params=[CHIP, WORKSPACE], ids=["chip", "workspace"])
def bsolver(request, solver):
    """both implementations of lqmpc_bounds_batch"""
    from lq_mpc_amd import KERNEL_GENERIC, KERNEL_AUTO
    solver.set_options(kernel=KERNEL_GENERIC if request.param == WORKSPACE else KERNEL_AUTO)
    solver.expected_bounds_kernel = request.param
    yield solver
    solver.set_options(kernel=KERNEL_AUTO)


def host_reference(N, A, B, Q, R, lb, ub, eA, eB, MV, x, p, V):
    F_u = np.vstack([np.diag(1.0 / ub), np.diag(1.0 / lb)])
    out = {k: [] for k in ("alpha", "beta", "xi", "eta", "bound", "eps", "K", "gamma", "rho_cl", "norm_Gamma", "norm_Phi", "min_eig_H")}
    for j in range(A.shape[2]):
        Aj, Bj = A[:, :, j], B[:, :, j]
        K = orc.dlqr_gain(Aj, Bj, Q, R)[0]
        try:
            ed = bounds_np.energy_decreasing(N, Aj, Bj, Q, R, F_u, eA[j], eB[j], -K, MV[j])
        except ValueError:                   # rho(A - BK) + 0.4 > 1 makes gamma negative: math domain error in the reference's formulas too
            ed = {"xi": np.nan, "eta": np.nan}
        eb = bounds_np.energy_bound(N, Aj, Bj, Q, R, lb, ub, eA[j], eB[j], x, p)
        st = bounds_np.stability_numbers(Aj, Bj, Q, R, -K)
        G, Phi = bounds_np.gamma_phi(N, Aj, Bj)
        hatH = np.kron(R, np.eye(N)) + G.T @ np.kron(Q, np.eye(N + 1)) @ G
        for k, v in (("alpha", eb["alpha"]), ("beta", eb["beta"]), ("xi", ed["xi"]), ("eta", ed["eta"]),
                     ("bound", (eb["alpha"] * V + eb["beta"]) / (1 - ed["xi"] - ed["eta"])),
                     ("eps", bounds_np.local_radius(F_u, -K, Q)), ("K", K), ("gamma", st["gamma"]),
                     ("rho_cl", np.max(np.abs(np.linalg.eigvals(Aj - Bj @ K)))), ("norm_Gamma", np.linalg.norm(G, 2)),
                     ("norm_Phi", np.linalg.norm(Phi, 2)), ("min_eig_H", np.min(np.linalg.eigvalsh(0.5 * (hatH + hatH.T))))):
            out[k].append(v)
    return {k: np.array(v) for k, v in out.items()}


def test_reference_systems_against_host_oracle_and_npz(bsolver, golden_dir):
    solver = bsolver
    """The reference's 1 000 perturbed systems at N = 7 (utils_class.py:802-859): every coefficient against the host oracle, and
    the four tables of the npz (M_V from the GPU's own max-V_N pass, as in data_generation)."""
    d = np.load(os.path.join(golden_dir, "data_lq_mpc_multipleSys.npz"))
    eA = np.load(os.path.join(golden_dir, "error_A_f.npy")); eB = np.load(os.path.join(golden_dir, "error_B_f.npy"))
    A = np.ascontiguousarray((A0[:, :, None, None] + eA).reshape(2, 2, 1000)); B = np.ascontiguousarray((B0[:, :, None, None] + eB).reshape(2, 1, 1000))
    lb, ub = np.array([-0.1]), np.array([0.1])
    r0 = solver.bounds_batch(7, A0[:, :, None], B0[:, :, None], Q2, R1, lb, ub, 0.0, 0.0, None, np.zeros(2), np.ones(3), 0.0)
    assert r0["status"][0] == 0
    np.testing.assert_allclose(r0["K"][:, :, 0], [[0.48363093, 0.45846723]], rtol=1e-7)
    assert abs(r0["eps"][0] - 0.04503580745099056) < 1e-14
    x0_vec = orc.circle_generator(8, 1.5, r0["eps"][0], Q2)
    xs = x0_vec[:, 1]
    V = float(d["V_expert"])
    MV = solver.max_vn_batch(7, A, B, Q2, R1, Q2, lb, ub, x0_vec)["M_V"]
    lev = np.tile(d["error"], 100)                                      # instance (j, i) has error level error[i]
    g = solver.bounds_batch(7, A, B, Q2, R1, lb, ub, lev, lev, MV, xs, P3, V, want_aux=True)
    assert solver.last_kernel() == solver.expected_bounds_kernel
    assert np.all(g["status"] == 0)
    for k, name in (("xi", "xi_table_error"), ("alpha", "alpha_table_error"), ("beta", "beta_table_error"), ("bound", "bound_table_error")):
        np.testing.assert_allclose(g[k].reshape(100, 10), d[name], rtol=1e-8, err_msg=name)
    idx = np.arange(0, 1000, 9)
    h = host_reference(7, A[:, :, idx], B[:, :, idx], Q2, R1, lb, ub, lev[idx], lev[idx], MV[idx], xs, P3, V)
    for k in ("alpha", "beta", "xi", "eta", "bound", "eps", "gamma", "rho_cl", "norm_Gamma", "norm_Phi", "min_eig_H"):
        assert rel(g[k][idx], h[k]) < 1e-9, k
    assert np.max(np.abs(g["K"][:, :, idx].transpose(2, 0, 1) - h["K"])) < 1e-10


@pytest.mark.parametrize("nx,nu,N,Bsz", [(4, 2, 10, 300), (3, 3, 5, 130), (8, 4, 12, 70), (2, 1, 30, 64), (5, 2, 1, 40)])
@pytest.mark.parametrize("weights", ["dense", "scalar_Q", "scalar"])
def test_random_systems_dense_weights(bsolver, nx, nu, N, Bsz, weights):
    solver = bsolver
    """Random stabilisable models, dense SPD Q and R (exercises the Kronecker ordering of hat H, utils.py:316-319), asymmetric box,
    per-instance error levels and energy bars."""
    rng = np.random.default_rng(10 * nx + nu + N)
    A = rng.standard_normal((nx, nx, Bsz))
    A *= rng.uniform(0.2, 0.9, Bsz) / np.abs(np.linalg.eigvals(A.transpose(2, 0, 1))).max(axis=1)
    B = rng.standard_normal((nx, nu, Bsz))
    def spd(m, lo, hi):
        M = rng.standard_normal((m, m)); M = M @ M.T / m + np.eye(m)
        return M * rng.uniform(lo, hi)
    Q, R = spd(nx, 0.5, 3.0), spd(nu, 0.2, 2.0)
    if weights != "dense":                   # Q = q I: hat H = kron(R, I) + q Gamma'Gamma; R = r I as well: no second eigenproblem on the chip
        Q = 1.7 * np.eye(nx)
    if weights == "scalar":
        R = 0.6 * np.eye(nu)
    lb, ub = -rng.uniform(0.05, 0.5, nu), rng.uniform(0.05, 0.5, nu)
    eA, eB = rng.uniform(1e-3, 1e-2, Bsz), rng.uniform(1e-3, 1e-2, Bsz)
    MV = rng.uniform(0.1, 5.0, Bsz)
    x, p, V = rng.standard_normal(nx) * 0.2, np.array([0.3, 1.5, 0.7]), 1.7
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    g = solver.bounds_batch(N, A, B, Q, R, lb, ub, eA, eB, MV, x, p, V, want_aux=True)
    assert solver.last_kernel() == solver.expected_bounds_kernel
    h = host_reference(N, A, B, Q, R, lb, ub, eA, eB, MV, x, p, V)
    # status 3 = non-finite coefficients: exactly the models for which the reference's formulas leave the reals
    # (gamma < 0 when rho(A - BK) + 0.4 > 1, utils.py:358-371)
    assert np.array_equal(g["status"] == 0, np.isfinite(h["xi"])) and np.mean(g["status"] == 0) > 0.5 and np.all((g["status"] == 0) | (g["status"] == 3))
    assert np.max(np.abs(g["K"].transpose(2, 0, 1) - h["K"])) < 1e-9 * max(1.0, np.abs(h["K"]).max())
    for k in ("eps", "rho_cl", "norm_Gamma", "norm_Phi", "min_eig_H", "alpha", "beta"):
        assert rel(g[k], h[k]) < 1e-9, k
    # gamma = C / (1 - (rho + 0.4)^2) has a pole at rho = 0.6: compare where it is well conditioned
    ok = (np.abs(1 - (h["rho_cl"] + 0.4) ** 2) > 1e-3) & (g["status"] == 0)
    assert ok.mean() > 0.4
    for k in ("gamma", "xi", "eta", "bound"):
        fin = ok & np.isfinite(h[k])
        assert rel(g[k][fin], h[k][fin]) < 1e-7, k


def test_unstabilisable_model_is_reported(bsolver):
    solver = bsolver
    A = np.zeros((2, 2, 70)); A[0, 0] = 1.5; A[1, 1] = 0.5
    B = np.zeros((2, 1, 70)); B[1, 0] = 1.0                        # the unstable mode is not reachable
    B[0, 0, 35:] = 1.0                                              # ... except in the second half of the batch
    g = solver.bounds_batch(5, A, B, Q2, R1, [-0.1], [0.1], 1e-3, 1e-3, 1.0, np.ones(2), np.ones(3), 1.0)
    assert np.all((g["status"][:35] == 1) | (g["status"][:35] == 2)) and np.all((g["status"][35:] == 0) | (g["status"][35:] == 3))
    K = orc.dlqr_gain(A[:, :, 40], B[:, :, 40], Q2, R1)[0]
    assert np.max(np.abs(g["K"][:, :, 40] - K)) < 1e-10
    with pytest.raises(Exception):
        solver.bounds_batch(5, A, B, Q2, -R1, [-0.1], [0.1], 1e-3, 1e-3, 1.0, np.ones(2), np.ones(3), 1.0)     # R not SPD
    with pytest.raises(Exception):
        solver.bounds_batch(5, A, B, Q2, R1, [0.0], [0.1], 1e-3, 1e-3, 1.0, np.ones(2), np.ones(3), 1.0)      # a zero bound is no row of F_u


def test_slowly_damped_true_system_keeps_its_radius(solver):
    """A true system whose LQR closed loop has rho(A - BK) + 0.4 > 1: the bound formulas leave the reals (status 3, utils.py:358)
    but K and the local radius are valid, and the reference computes local_radius for any stabilisable system
    (utils_class.py:761-764): epsilon_lqr and circle_generator must work."""
    from lq_mpc_amd.sweep import LQ_RDP_Behavior_Multiple, circle_generator
    A = np.array([[1.02, 0.3], [0.0, 0.97]]); B = np.array([[0.05], [0.1]])
    Q, R = 0.02 * np.eye(2), 4.0 * np.eye(1)
    K, _ = orc.dlqr_gain(A, B, Q, R)
    rho = np.max(np.abs(np.linalg.eigvals(A - B @ K)))
    assert 0.6 < rho < 1.0
    beh = LQ_RDP_Behavior_Multiple.__new__(LQ_RDP_Behavior_Multiple)
    beh.A_true, beh.B_true, beh.Q, beh.R = A, B, Q, R
    beh.lb, beh.ub = np.array([-0.1]), np.array([0.1])
    beh._solver, beh._eps_lqr = solver, None
    eps = beh.epsilon_lqr
    F_u = np.array([[10.0], [-10.0]])
    assert abs(eps - orc.local_radius(F_u, -K, Q)) <= 1e-9 * eps
    assert np.max(np.abs(beh.K_lqr - K)) <= 1e-9 * np.max(np.abs(K))
    pts = circle_generator(8, 1.5, eps, Q)
    assert pts.shape == (2, 8) and np.all(np.isfinite(pts))


def test_c5_shape_on_chip_against_workspace_kernel_and_host(solver):
    """n_x = 8, n_u = 4, N = 30 (C5: n = 120, one system per wavefront, 116 KB of LDS, compiled at run time): the on-chip kernels
    against the HBM-workspace kernel on every output, and against the host oracle on a few systems."""
    from lq_mpc_amd import synth, KERNEL_GENERIC, KERNEL_AUTO
    b = synth.make_batch(5, Bsz=40)
    N, Bsz = b["N"], 40
    rng = np.random.default_rng(5)
    eA, eB, MV = rng.uniform(1e-3, 1e-2, Bsz), rng.uniform(1e-3, 1e-2, Bsz), rng.uniform(0.1, 2.0, Bsz)
    x, p, V = b["x0"][:, 0].copy(), P3, 1.3
    a = (N, b["A"], b["B"], b["Q"], b["R"], b["lb"], b["ub"], eA, eB, MV, x, p, V)
    try:
        g = solver.bounds_batch(*a, want_aux=True)
        assert solver.last_kernel() == CHIP
        solver.set_options(kernel=KERNEL_GENERIC)
        w = solver.bounds_batch(*a, want_aux=True)
        assert solver.last_kernel() == WORKSPACE
    finally:
        solver.set_options(kernel=KERNEL_AUTO)
    assert np.array_equal(g["status"], w["status"])
    for k in ("eps", "rho_cl", "norm_A", "norm_B", "norm_K", "norm_Gamma", "norm_Phi", "min_eig_H", "alpha", "beta"):
        assert rel(g[k], w[k]) < 1e-9, k
    assert np.max(np.abs(g["K"] - w["K"])) < 1e-9 * np.abs(w["K"]).max()
    idx = np.arange(0, Bsz, 13)
    h = host_reference(N, b["A"][:, :, idx], b["B"][:, :, idx], b["Q"], b["R"], b["lb"], b["ub"], eA[idx], eB[idx], MV[idx], x, p, V)
    for k in ("eps", "rho_cl", "norm_Gamma", "norm_Phi", "min_eig_H", "alpha", "beta"):
        assert rel(g[k][idx], h[k]) < 1e-8, k
