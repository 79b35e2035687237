"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the golden data.

Tolerances (BASELINE.json north_star: "match ... to rtol 1e-5"):
  V_N, J_T, M_V : relative error <= RTOL = 1e-5
  u_0, U        : |u - u*| <= RTOL * max(|u*|, 1e-3 * u_max)   (late in a rollout u -> 0)
The kernels normally land 6+ orders of magnitude inside these; TIGHT guards that margin.
"""
import os

import numpy as np
import pytest

from lq_mpc_amd import BatchSolver, LQ_MPC_Controller, LQ_MPC_Simulator, synth, KERNEL_GENERIC, KERNEL_AUTO
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-5
TIGHT = 1e-8
U_MAX = 0.1

A0 = np.array([[1.0, 0.7], [0.12, 0.4]])
B0 = np.array([[1.0], [1.2]])
Q2 = 2.0 * np.eye(2)
R1 = np.eye(1)
F_U = np.vstack((10 * np.eye(1), -10 * np.eye(1)))
X_START = np.array([0.15916231240837822, 0.15916231240837819])


def rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


def u_err(u, ur, umax=U_MAX):
    return np.max(np.abs(u - ur) / np.maximum(np.abs(ur), 1e-3 * umax))


def args(b):
    return (b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["lb"], b["ub"])


KERNELS = [KERNEL_GENERIC, KERNEL_AUTO]


@pytest.fixture(params=KERNELS, ids=["generic", "auto"])
def ksolver(request, solver):
    solver.set_options(kernel=request.param)
    yield solver
    solver.set_options(kernel=KERNEL_AUTO)


# ---------------- seeded synthetic configs vs the oracle ----------------
@pytest.mark.parametrize("cfg,bsz", [(1, 1), (2, 512), (3, 512), (4, 128), (5, 64)])
def test_solve_batch_vs_oracle(ksolver, cfg, bsz, golden_dir):
    b = synth.make_batch(cfg, Bsz=bsz, fixture_dir=golden_dir)
    got = ksolver.solve_batch(*args(b), b["x0"])
    ref = orc.solve_batch(*args(b), b["x0"])
    assert np.all(got["status"] == 0)
    assert rel(got["V_N"], ref["V_N"]) < RTOL and u_err(got["u_0"], ref["u_0"]) < RTOL
    assert rel(got["V_N"], ref["V_N"]) < TIGHT and u_err(got["u_0"], ref["u_0"]) < 1e-6


@pytest.mark.parametrize("cfg,bsz,T", [(1, 1, 50), (2, 512, 30), (3, 512, 30), (4, 64, 10), (5, 64, 4)])
def test_rollout_batch_vs_oracle(ksolver, cfg, bsz, T, golden_dir):
    b = synth.make_batch(cfg, Bsz=bsz, fixture_dir=golden_dir)
    got = ksolver.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    ref = orc.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    assert np.all(got["status"] == 0)
    assert rel(got["J_T"], ref["J_T"]) < RTOL and rel(got["J_T"], ref["J_T"]) < TIGHT
    assert u_err(got["U"], ref["U"]) < RTOL
    scale = np.max(np.abs(ref["X"]))
    assert np.max(np.abs(got["X"] - ref["X"])) < RTOL * scale
    assert np.all(got["iters"] >= 0) and got["iters"].sum() > 0


@pytest.mark.parametrize("cfg,bsz", [(2, 256), (3, 128)])
def test_max_vn_batch_vs_oracle(ksolver, cfg, bsz, golden_dir):
    b = synth.make_batch(cfg, Bsz=bsz, fixture_dir=golden_dir)
    rng = np.random.default_rng(5)
    x0s = rng.standard_normal((b["nx"], 8)) * 0.3
    got = ksolver.max_vn_batch(*args(b), x0s)
    ref = orc.max_vn_batch(*args(b), x0s)
    assert np.all(got["status"] == 0) and rel(got["M_V"], ref) < TIGHT


# ---------------- the reference's own fixture data through the GPU path ----------------
def test_golden_true_cost_tables(ksolver, golden_dir):
    """error_{A,B}_f.npy -> true_cost_error / true_cost_horizon of the reference's npz (utils_class.py:802-833, 886-916)."""
    d = np.load(os.path.join(golden_dir, "data_lq_mpc_multipleSys.npz"))
    eA = np.load(os.path.join(golden_dir, "error_A_f.npy"))
    eB = np.load(os.path.join(golden_dir, "error_B_f.npy"))
    lb, ub = np.array([-0.1]), np.array([0.1])
    A = (A0[:, :, None, None] + eA).reshape(2, 2, 1000)      # the file layout IS the SoA layout
    B = (B0[:, :, None, None] + eB).reshape(2, 1, 1000)
    x0 = np.repeat(X_START[:, None], 1000, 1)
    J = ksolver.rollout_batch(30, 7, A, B, Q2, R1, Q2, lb, ub, x0, A0, B0)["J_T"].reshape(100, 10)
    assert rel(J, d["true_cost_error"]) < RTOL and rel(J, d["true_cost_error"]) < 1e-10
    for k, N in enumerate(range(6, 11)):
        A4, B4 = A0[:, :, None] + eA[:, :, :, 4], B0[:, :, None] + eB[:, :, :, 4]
        J = ksolver.rollout_batch(30, N, A4, B4, Q2, R1, Q2, lb, ub, x0[:, :100], A0, B0)["J_T"]
        assert rel(J, d["true_cost_horizon"][:, k]) < 1e-10
    V = ksolver.solve_batch(30, A0[:, :, None], B0[:, :, None], Q2, R1, Q2, lb, ub, X_START[:, None])["V_N"][0]
    assert abs(V - float(d["V_expert"])) / float(d["V_expert"]) < 1e-10


def test_dropin_classes_reproduce_reference_scripts(solver):
    """mpc_test.py:33-66 and working_example_single.py:62-66 with the reference's class API."""
    x0 = np.array([0.1125, 0.19])
    mpc = LQ_MPC_Controller(20, A0, B0, Q2, R1, Q2, F_U)
    info = mpc.solve(x0, np.zeros((2, 20)), np.zeros((1, 20)))
    assert set(info) == {"u_0", "V_N"} and info["u_0"].shape == (1,) and isinstance(info["V_N"], float)
    assert abs(info["V_N"] - 0.17375571642447996) < 1e-10 and abs(info["u_0"][0] + 0.1) < 1e-9
    sim = LQ_MPC_Simulator(20, 6, A0, B0, Q2, R1, Q2, F_U)
    tr = sim.simulate(x0, np.array([[1.01, 0.7], [0.12, 0.41]]), np.array([[1.0], [1.21]]), np.zeros((2, 6)), np.zeros((1, 6)))
    assert set(tr) == {"X", "U", "J_T"} and tr["X"].shape == (2, 21) and tr["U"].shape == (1, 20)
    assert abs(tr["J_T"] - 0.17571889572646185) < 1e-10
    np.testing.assert_allclose(tr["X"][:, 0], x0)
    ref = orc.simulate(20, 6, A0, B0, Q2, R1, Q2, [-0.1], [0.1], x0, np.array([[1.01, 0.7], [0.12, 0.41]]), np.array([[1.0], [1.21]]))
    assert np.max(np.abs(tr["X"] - ref["X"])) < 1e-9 and np.max(np.abs(tr["U"] - ref["U"])) < 1e-9
    x0_vec = orc.circle_generator(8, 1.5, 0.04503580745099056, Q2)
    mv = max(LQ_MPC_Controller(6, A0, B0, Q2, R1, Q2, F_U).solve(x0_vec[:, k], np.zeros((2, 6)), np.zeros((1, 6)))["V_N"] for k in range(8))
    assert abs(mv - 0.2022946791688417) < 1e-10


# ---------------- edge cases ----------------
def test_zero_state_and_saturated_cases(ksolver):
    b = synth.make_batch(3, Bsz=64)
    x0 = b["x0"].copy()
    x0[:, :16] = 0.0                      # exactly at the origin: u = 0, V = 0
    x0[:, 16:32] *= 50.0                  # far outside: every early input saturated
    x0[:, 32:48] *= 1e-6                  # tiny: unconstrained, u linear in x0
    got = ksolver.solve_batch(*args(b), x0)
    ref = orc.solve_batch(*args(b), x0)
    assert np.all(got["status"] == 0)
    assert np.all(np.abs(got["u_0"][:, :16]) < 1e-100) and np.all(np.abs(got["V_N"][:16]) < 1e-100)
    assert np.all(np.abs(np.abs(got["u_0"][:, 16:32]).max(0) - U_MAX) < 1e-12)
    assert u_err(got["u_0"][:, 16:], ref["u_0"][:, 16:]) < RTOL
    assert rel(got["V_N"][16:], ref["V_N"][16:]) < TIGHT
    np.testing.assert_allclose(got["u_0"][:, 32:48], ref["u_0"][:, 32:48], rtol=1e-9)


def test_references_terminal_weight_asymmetric_box_and_per_instance_plant(ksolver):
    rng = np.random.default_rng(3)
    nx, nu, N, Bsz, T = 3, 2, 6, 96, 8
    A = np.ascontiguousarray(0.7 * rng.standard_normal((nx, nx, 1)) + 0.05 * rng.standard_normal((nx, nx, Bsz)))
    B = np.ascontiguousarray(rng.standard_normal((nx, nu, 1)) + 0.05 * rng.standard_normal((nx, nu, Bsz)))
    Q, R, P = np.diag([2.0, 1.0, 3.0]), np.array([[1.0, 0.2], [0.2, 0.5]]), np.diag([5.0, 4.0, 1.0])
    lb, ub = np.array([-0.3, -0.05]), np.array([0.1, 0.4])
    x0 = rng.standard_normal((nx, Bsz))
    xr, ur = 0.3 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N))
    At = np.ascontiguousarray(A + 0.01 * rng.standard_normal((nx, nx, Bsz)))
    Bt = np.ascontiguousarray(B + 0.01 * rng.standard_normal((nx, nu, Bsz)))
    got = ksolver.solve_batch(N, A, B, Q, R, P, lb, ub, x0, xr, ur)
    ref = orc.solve_batch(N, A, B, Q, R, P, lb, ub, x0, xr, ur)
    assert np.all(got["status"] == 0)
    assert rel(got["V_N"], ref["V_N"]) < TIGHT and u_err(got["u_0"], ref["u_0"], 0.1) < RTOL
    g2 = ksolver.rollout_batch(T, N, A, B, Q, R, P, lb, ub, x0, At, Bt, xr, ur, want_traj=True)
    r2 = orc.rollout_batch(T, N, A, B, Q, R, P, lb, ub, x0, At, Bt, xr, ur, want_traj=True)
    assert rel(g2["J_T"], r2["J_T"]) < TIGHT and u_err(g2["U"], r2["U"], 0.1) < RTOL


def test_cold_start_guess_from_the_saturated_roll(solver):
    """The 16-lane-row kernels start the first QP of an instance from the face a clipped roll-forward under the stage gains predicts
    (lqmpc_r16_setup.h, ROLL) instead of the rows of the unconstrained minimiser outside the box: same optimum, fewer active-set
    iterations (C3 default mix 2.19 -> 1.41 per QP, C2 hard mix 6.5 -> 1.0).  With references or an off-centre box the guess is
    not used and the count stays what it was."""
    for cfg, mix, cap in ((3, "default", 1.6), (3, "hard", 3.0), (2, "default", 1.2), (2, "hard", 1.5)):
        b = synth.make_batch(cfg, Bsz=2048, mix=mix)
        got = solver.solve_batch(*args(b), b["x0"])
        ref = orc.solve_batch(*args(b), b["x0"])
        assert "r16" in solver.last_kernel() and np.all(got["status"] == 0)
        assert rel(got["V_N"], ref["V_N"]) < TIGHT and u_err(got["u_0"], ref["u_0"]) < RTOL
        assert got["iters"].mean() < cap, (cfg, mix, got["iters"].mean())
        # an off-centre box (has_lin): the plain cold start, and still the optimum
        lb2, ub2 = b["lb"] - 0.02, b["ub"] - 0.02
        g2 = solver.solve_batch(b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], lb2, ub2, b["x0"])
        r2 = orc.solve_batch(b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], lb2, ub2, b["x0"])
        assert np.all(g2["status"] == 0) and rel(g2["V_N"], r2["V_N"]) < TIGHT and u_err(g2["u_0"], r2["u_0"]) < RTOL


def test_ragged_batch_sizes(ksolver):
    """Batches that do not fill a wavefront / workgroup, including a single instance."""
    for bsz in (1, 3, 63, 65, 257):
        b = synth.make_batch(3, Bsz=bsz)
        got = ksolver.rollout_batch(5, *args(b), b["x0"], b["A_true"], b["B_true"])
        ref = orc.rollout_batch(5, *args(b), b["x0"], b["A_true"], b["B_true"])
        assert got["J_T"].shape == (bsz,) and rel(got["J_T"], ref["J_T"]) < TIGHT


def test_bad_arguments_are_reported(solver):
    b = synth.make_batch(2, Bsz=8)
    with pytest.raises(Exception):
        solver.solve_batch(b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["ub"], b["lb"], b["x0"])   # empty box
    with pytest.raises(Exception):
        solver.rollout_batch(0, *args(b), b["x0"], b["A_true"], b["B_true"])                            # T = 0
    with pytest.raises(ValueError):
        LQ_MPC_Controller(5, A0, B0, Q2, R1, Q2, np.array([[1.0], [2.0]]))                               # one-sided


# ---------------- size-independent properties at BASELINE.json's full sizes ----------------
@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_full_size_properties(solver, cfg, golden_dir):
    """Odd symmetry (zero refs, symmetric box): solve(-x0) = (-u_0, V_N); bounds respected; x0 -> 0 gives the
    unconstrained linear law; a 1024-instance sample agrees with the oracle."""
    b = synth.make_batch(cfg, fixture_dir=golden_dir)
    assert b["Bsz"] == synth.CONFIGS[cfg]["Bsz"]
    g1 = solver.solve_batch(*args(b), b["x0"])
    g2 = solver.solve_batch(*args(b), -b["x0"])
    assert np.all(g1["status"] == 0) and np.all(g2["status"] == 0)
    assert np.max(np.abs(g1["u_0"] + g2["u_0"])) < 1e-9 and rel(g1["V_N"], g2["V_N"]) < 1e-10
    assert np.all(np.abs(g1["u_0"]) <= U_MAX + 1e-15)
    s1 = solver.solve_batch(*args(b), 1e-4 * b["x0"])
    s2 = solver.solve_batch(*args(b), 2e-4 * b["x0"])
    assert np.max(np.abs(2 * s1["u_0"] - s2["u_0"])) < 1e-12 and rel(4 * s1["V_N"], s2["V_N"]) < 1e-9
    idx = np.random.default_rng(0).choice(b["Bsz"], 1024 if cfg < 5 else 256, replace=False)
    sub = dict(b, A=np.ascontiguousarray(b["A"][:, :, idx]), B=np.ascontiguousarray(b["B"][:, :, idx]))
    ref = orc.solve_batch(*args(sub), np.ascontiguousarray(b["x0"][:, idx]))
    assert rel(g1["V_N"][idx], ref["V_N"]) < TIGHT and u_err(g1["u_0"][:, idx], ref["u_0"]) < RTOL
    # the full-size rollout of every config (C5: the workgroup kernel, ~30 ms per launch), a sample against the oracle
    r1 = solver.rollout_batch(30, *args(b), b["x0"], b["A_true"], b["B_true"])
    if cfg == 5:
        assert solver.last_kernel() == "lqmpc_wg_kernel"
        idx = idx[:96]
        sub = dict(b, A=np.ascontiguousarray(b["A"][:, :, idx]), B=np.ascontiguousarray(b["B"][:, :, idx]))
    rr = orc.rollout_batch(30, *args(sub), np.ascontiguousarray(b["x0"][:, idx]), b["A_true"], b["B_true"])
    assert np.all(r1["status"] == 0) and rel(r1["J_T"][idx], rr["J_T"]) < TIGHT
    # odd symmetry of the closed loop: the rollout from -x0 costs the same
    r2 = solver.rollout_batch(30, *args(b), -b["x0"], b["A_true"], b["B_true"])
    assert rel(r1["J_T"], r2["J_T"]) < 1e-9


@pytest.mark.parametrize("cfg,bsz", [(2, 1024), (3, 2048), (3, 12288), (4, 512), (5, 48)])
def test_hard_mix_rollout_vs_oracle(solver, cfg, bsz, golden_dir):
    """The hard initial-state mix of bench.py's `rollout_hard` leg (synth.make_batch(mix="hard"): x0 6-24x outside the region
    where the box is inactive): most MPC steps are constrained QPs, so this is the active-set machinery, not the presolve.
    12 288 instances of C3 also take the sorted walk (probe + bucket order)."""
    b = synth.make_batch(cfg, Bsz=bsz, fixture_dir=golden_dir, mix="hard")
    T = 30 if cfg < 5 else 8
    got = solver.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    ref = orc.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    assert np.all(got["status"] == 0)
    assert rel(got["J_T"], ref["J_T"]) < TIGHT and u_err(got["U"], ref["U"]) < RTOL
    assert np.max(np.abs(got["X"] - ref["X"])) < 1e-7 * np.max(np.abs(ref["X"]))
    m = min(bsz, 128)
    share = synth.constrained_share(b, ref["X"][:, :, :m], np.arange(m))
    floor = 0.5 if cfg == 3 else 0.3                 # C3 is the bench leg: most of its steps must be constrained
    assert share > floor, f"hard mix: only {share:.2f} of the steps are constrained"
    assert got["iters"].sum() > 0.5 * bsz * T * floor          # at least one factorisation per constrained step, roughly


def test_sweep_batch_on_c3_shapes_with_level_set_points(solver):
    """SURVEY 8(f) rank 1 end to end at n_x = 4: circle_generator's level-set points (seeded planes) feed lqmpc_sweep_batch --
    M_V over the 8 points and the rollout in one launch -- against the oracle's max_vn + rollout."""
    from lq_mpc_amd import sweep as sw
    b = synth.make_batch(3, Bsz=3000)
    base = float(np.median(np.einsum("ib,ij,jb->b", b["x0"], b["Q"], b["x0"])))
    x0s = sw.circle_generator(8, 1.5, base, b["Q"])
    assert x0s.shape == (4, 8)
    np.testing.assert_allclose(np.einsum("ik,ij,jk->k", x0s, b["Q"], x0s), 1.5 ** 2 * base, rtol=1e-12)
    g = solver.sweep_batch(30, *args(b), b["x0"], x0s, b["A_true"], b["B_true"])
    assert "r16" in solver.last_kernel() and np.all(g["status"] == 0)
    mv = orc.max_vn_batch(*args(b), x0s)
    jt = orc.rollout_batch(30, *args(b), b["x0"], b["A_true"], b["B_true"])["J_T"]
    assert rel(g["M_V"], mv) < TIGHT and rel(g["J_T"], jt) < TIGHT
    assert np.mean(g["M_V"] > 1.0001 * x0s[:, 0] @ b["Q"] @ x0s[:, 0]) > 0.99      # V_N includes x0'Qx0 and then some


# ---------------- solver options ----------------
@pytest.mark.parametrize("cfg", [2, 3])
def test_presolve_and_polish_options_agree(solver, cfg, golden_dir):
    """presolve (unconstrained-minimiser shortcut) on/off and polish on/off give the same answers."""
    b = synth.make_batch(cfg, Bsz=1024, fixture_dir=golden_dir)
    ref = orc.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    r1 = orc.solve_batch(*args(b), b["x0"])
    try:
        for presolve in (0, 1):
            solver.set_options(presolve=presolve, polish=1, warm_start=0)
            got = solver.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
            assert np.all(got["status"] == 0)
            assert rel(got["J_T"], ref["J_T"]) < TIGHT and u_err(got["U"], ref["U"]) < RTOL
            g1 = solver.solve_batch(*args(b), b["x0"])
            assert rel(g1["V_N"], r1["V_N"]) < TIGHT and u_err(g1["u_0"], r1["u_0"]) < RTOL
            if presolve:
                assert got["iters"].sum() < iters_off.sum()       # interior steps cost no iterations
            else:
                iters_off = got["iters"]
        solver.set_options(presolve=-1, polish=0, eps=1e-13, warm_start=0)       # interior point only: still inside RTOL
        got = solver.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
        assert rel(got["J_T"], ref["J_T"]) < RTOL and u_err(got["U"], ref["U"]) < RTOL
        for presolve in (0, 1):                                    # active-set warm start with / without presolve
            solver.set_options(presolve=presolve, polish=1, eps=1e-12, warm_start=1)
            got = solver.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
            assert np.all(got["status"] == 0)
            assert rel(got["J_T"], ref["J_T"]) < TIGHT and u_err(got["U"], ref["U"]) < RTOL
            g1 = solver.solve_batch(*args(b), b["x0"])
            assert rel(g1["V_N"], r1["V_N"]) < TIGHT and u_err(g1["u_0"], r1["u_0"]) < RTOL
    finally:
        solver.set_options(presolve=-1, polish=1, eps=1e-12, warm_start=-1)


# ---------------- the batch caller: data_generation's hot half ----------------
def test_data_generation_reproduces_reference_npz(solver, golden_dir):
    """working_example_multiple.py:13-58, 70-76 constants -> utils_class.py:766-833, 861-916."""
    from lq_mpc_amd.sweep import LQ_RDP_Behavior_Multiple
    info_opc = {"A": A0, "B": B0, "Q": Q2, "R": R1, "F_u": F_U}
    info_N = {"N_min": 6, "N_max": 10, "N_nominal": 7, "N_opc": 30, "N_mpc": 30}
    info_e = {"e_min": 1e-3, "e_max": 1e-2, "e_nominal": 5e-3}
    info_ref = {"x_ref": np.zeros((2, 7)), "u_ref": np.zeros((1, 7)), "x_ref_long": np.zeros((2, 30)), "u_ref_long": np.zeros((1, 30))}
    beh = LQ_RDP_Behavior_Multiple(info_opc, info_N, info_e, 20, "f", data_dir=golden_dir, solver=solver)
    out = beh.data_generation(8, 1.5, info_ref, np.array([0.1, 1, 0.6]))
    seq = beh.data_generation(8, 1.5, info_ref, np.array([0.1, 1, 0.6]), concurrent=False)      # the six passes on one handle, in turn
    for k, v in out.items():
        assert np.array_equal(np.asarray(v), np.asarray(seq[k])), k
    d = np.load(os.path.join(golden_dir, "data_lq_mpc_multipleSys.npz"))
    np.testing.assert_allclose(out["error"], d["error"]); np.testing.assert_array_equal(out["horizon"], d["horizon"])
    assert abs(out["V_expert"] - float(d["V_expert"])) / float(d["V_expert"]) < 1e-10
    assert rel(out["true_cost_error"], d["true_cost_error"]) < 1e-10
    assert rel(out["true_cost_horizon"], d["true_cost_horizon"]) < 1e-10
    # all 13 arrays of the reference's npz: the bound coefficients are fed with the GPU's M_V and V_expert
    assert set(d.files) <= set(out)
    for k in ("xi_table_error", "xi_table_horizon", "alpha_table_error", "alpha_table_horizon",
              "beta_table_error", "beta_table_horizon", "bound_table_error", "bound_table_horizon"):
        np.testing.assert_allclose(out[k], d[k], rtol=1e-8, err_msg=k)
    # M_V is not in the npz (only through xi); pin it against the oracle
    eA = np.load(os.path.join(golden_dir, "error_A_f.npy")); eB = np.load(os.path.join(golden_dir, "error_B_f.npy"))
    A = (A0[:, :, None, None] + eA).reshape(2, 2, 1000); B = (B0[:, :, None, None] + eB).reshape(2, 1, 1000)
    mv = orc.max_vn_batch(7, A, B, Q2, R1, Q2, [-0.1], [0.1], out["x0_vec"]).reshape(100, 10)
    assert rel(out["M_V_error"], mv) < TIGHT
    assert np.all(out["M_V_error"] / beh.epsilon_lqr > 4.0) and np.all(out["M_V_error"] / beh.epsilon_lqr < 5.0)


def test_difficulty_ordering_is_transparent(solver, golden_dir):
    """options.order = 1 (probe + bucket sort on the logarithm of the key, hardest first)
    must not change any instance's result beyond the rounding of a different summation order."""
    b = synth.make_batch(3, Bsz=4096 + 37)
    try:
        solver.set_options(order=0)
        r0 = solver.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
        solver.set_options(order=1)
        r1 = solver.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
        solver.set_options(order=1, presolve=0, warm_start=0)
        r2 = solver.rollout_batch(20, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    finally:
        solver.set_options(order=-1, presolve=-1, warm_start=-1)
    for k in ("J_T", "X", "U"):                                # same algorithm and iterates; the packed tier and its wide tier sum in different orders
        assert np.abs(r0[k] - r1[k]).max() <= 1e-12 * max(1.0, np.abs(r0[k]).max())
    packed = np.abs(r0["J_T"] - r1["J_T"]) == 0.0
    assert packed.mean() > 0.9                                # everything outside the wide tier is bit-identical
    # without presolve the interior steps go through the interior-point loop instead: same answers to tolerance
    assert rel(r0["J_T"], r2["J_T"]) < TIGHT and u_err(r0["U"], r2["U"]) < RTOL
    assert np.mean(r0["iters"] != r1["iters"]) < 1e-3
    assert np.all(r1["status"] == 0) and np.all(r2["status"] == 0)
    idx = np.random.default_rng(1).choice(b["Bsz"], 512, replace=False)
    sub = dict(b, A=np.ascontiguousarray(b["A"][:, :, idx]), B=np.ascontiguousarray(b["B"][:, :, idx]))
    ref = orc.rollout_batch(20, *args(sub), np.ascontiguousarray(b["x0"][:, idx]), b["A_true"], b["B_true"])
    assert rel(r1["J_T"][idx], ref["J_T"]) < TIGHT


def test_both_order_keys_are_transparent(solver):
    """The order has two keys (lqmpc_probe.h): the clipped roll of the shared plant (zero references, centred box) and the
    free-response gradient for everything else (per-instance plants, an off-centre box, T beyond the roll's 31 steps is capped).
    Whatever the key, ordering on / off gives the same rollouts; checked on the 16-lane-row kernel and against the oracle."""
    b = synth.make_batch(3, Bsz=2048 + 5)
    rng = np.random.default_rng(11)
    At = np.ascontiguousarray(b["A_true"][:, :, None] + 1e-3 * rng.standard_normal((4, 4, b["Bsz"])))
    Bt = np.ascontiguousarray(b["B_true"][:, :, None] + 1e-3 * rng.standard_normal((4, 2, b["Bsz"])))
    cases = {
        "shared plant": (b["lb"], b["ub"], b["A_true"], b["B_true"], 20),
        "shared plant, T = 40": (b["lb"], b["ub"], b["A_true"], b["B_true"], 40),
        "per-instance plants": (b["lb"], b["ub"], At, Bt, 12),
        "off-centre box": (b["lb"] - 0.03, b["ub"] - 0.03, b["A_true"], b["B_true"], 12),
    }
    idx = rng.choice(b["Bsz"], 256, replace=False)
    try:
        for name, (lb, ub, A_t, B_t, T) in cases.items():
            a = (b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], lb, ub)
            solver.set_options(order=0)
            r0 = solver.rollout_batch(T, *a, b["x0"], A_t, B_t)
            solver.set_options(order=1)
            r1 = solver.rollout_batch(T, *a, b["x0"], A_t, B_t)
            assert "r16" in solver.last_kernel(), name
            assert np.all(r0["status"] == 0) and np.all(r1["status"] == 0), name
            assert np.array_equal(r0["J_T"], r1["J_T"]) and np.array_equal(r0["iters"], r1["iters"]), name
            per = A_t.ndim == 3
            ref = orc.rollout_batch(T, b["N"], np.ascontiguousarray(b["A"][:, :, idx]), np.ascontiguousarray(b["B"][:, :, idx]), b["Q"], b["R"], b["P"],
                                    lb, ub, np.ascontiguousarray(b["x0"][:, idx]),
                                    np.ascontiguousarray(A_t[:, :, idx]) if per else A_t, np.ascontiguousarray(B_t[:, :, idx]) if per else B_t)
            assert rel(r1["J_T"][idx], ref["J_T"]) < TIGHT, name
    finally:
        solver.set_options(order=-1)


# ---------------- stress: random shapes of cost, box, conditioning on the specialised shapes ----------------
WG_SHAPES = [(8, 4, 30), (6, 3, 15), (16, 2, 17), (3, 2, 64), (5, 1, 33)]     # n = 120, 45, 34, 128, 33
PREBUILT = [(4, 2, 10), (2, 1, 10), (2, 1, 5), (2, 1, 7), (2, 1, 20), (2, 1, 30), (4, 2, 20)]
JIT_MAX_NX, JIT_MAX_NU = 8, 4              # lqmpc_r16_setup.h: SETUP_MAX_NX / SETUP_MAX_NU


def jit_domain(nx, nu, N):
    """Shapes the run-time compiled 16-lane-row kernel serves (lqmpc_jit.hip: jit_r16_shape)."""
    return nx <= JIT_MAX_NX and nu <= JIT_MAX_NU and N * nu <= 48


@pytest.mark.parametrize("nx,nu,N", [(4, 2, 10), (2, 1, 10), (2, 1, 5), (2, 1, 7), (2, 1, 20), (2, 1, 30), (3, 2, 6), (1, 1, 1), (5, 3, 4), (4, 2, 20)] + WG_SHAPES)
@pytest.mark.parametrize("warm", [0, 1])
def test_random_problems(solver, nx, nu, N, warm):
    """Unstable / badly scaled models, dense Q/R/P, asymmetric boxes, references, per-instance plants, on the
    specialised shapes (warm start on / off), on shapes only the generic kernel covers, and on the
    one-instance-per-workgroup shapes (32 < n <= 128, including stages that straddle the 16-row blocks)."""
    rng = np.random.default_rng(100 * nx + N + warm)
    wg = (nx, nu, N) in WG_SHAPES and not jit_domain(nx, nu, N)
    Bsz, T = (96, 6) if (nx, nu, N) in WG_SHAPES else (768, 12)
    A = rng.standard_normal((nx, nx, Bsz))
    rho_hi = 1.3 if N <= 20 else 200.0 ** (1.0 / N)           # keeps rho^N (the conditioning of the condensed Hessian) bounded
    A *= rng.uniform(0.3, rho_hi, Bsz) / np.abs(np.linalg.eigvals(A.transpose(2, 0, 1))).max(axis=1)   # spectral radius in [0.3, rho_hi]
    B = rng.standard_normal((nx, nu, Bsz)) * rng.uniform(0.1, 2.0, (1, 1, Bsz))
    def spd(m, lo, hi):
        M = rng.standard_normal((m, m)); M = M @ M.T / m + np.eye(m)
        return M * rng.uniform(lo, hi)
    Q, R, P = spd(nx, 0.5, 5.0), spd(nu, 0.05, 2.0), spd(nx, 0.5, 20.0)
    lb, ub = -rng.uniform(0.05, 0.5, nu), rng.uniform(0.05, 0.5, nu)
    x0 = rng.standard_normal((nx, Bsz)) * rng.choice([1e-3, 0.1, 1.0, 10.0], Bsz)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.05 * rng.standard_normal((nu, N))
    At = np.ascontiguousarray(A * 0.9 + 0.02 * rng.standard_normal((nx, nx, Bsz)))
    Bt = np.ascontiguousarray(B + 0.02 * rng.standard_normal((nx, nu, Bsz)))
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    umax = float(np.max(np.maximum(-lb, ub)))
    r1 = orc.solve_batch(N, A, B, Q, R, P, lb, ub, x0, xr, ur)
    r2 = orc.rollout_batch(T, N, A, B, Q, R, P, lb, ub, x0, At, Bt, xr, ur, want_traj=True)
    ok = np.isfinite(r2["J_T"]) & (np.abs(r2["X"]).max(axis=(0, 1)) < 1e6)     # diverging plants amplify round-off
    assert ok.mean() > 0.5
    # the two builds of the 16-lane-row kernel (one wave per SIMD for small batches, two for large): both, where both exist
    builds = (1, 0) if warm and (nx, nu, N) in [(4, 2, 10), (2, 1, 20)] else (-1,)
    for build in builds:
        try:
            solver.set_options(warm_start=warm, presolve=warm, r16_build=build)
            g1 = solver.solve_batch(N, A, B, Q, R, P, lb, ub, x0, xr, ur)
            g2 = solver.rollout_batch(T, N, A, B, Q, R, P, lb, ub, x0, At, Bt, xr, ur, want_traj=True)
            k = solver.last_kernel()                                  # packed or 16-lane-row specialisation (small batches, warm start on)
            jit = warm == 1 and (nx, nu, N) not in PREBUILT and jit_domain(nx, nu, N)      # compiled at run time (presolve + warm start: its algorithm)
            assert ("spec" in k or "r16" in k or "r64" in k) == ((nx, nu, N) in PREBUILT or jit)
            assert ("jit" in k) == jit
            assert ("wg" in k) == (wg or ((nx, nu, N) in WG_SHAPES and not jit))
        finally:
            solver.set_options(warm_start=-1, presolve=-1, r16_build=-1)
        assert np.all(g1["status"] == 0)
        assert rel(g1["V_N"], r1["V_N"]) < 1e-7 and u_err(g1["u_0"], r1["u_0"], umax) < RTOL
        assert np.all(g2["status"][ok] == 0)
        assert rel(g2["J_T"][ok], r2["J_T"][ok]) < 1e-6
        assert u_err(g2["U"][:, :3, ok], r2["U"][:, :3, ok], umax) < RTOL


def test_status_reports_iteration_cap(solver):
    """With the warm start off and a one-iteration budget the interior-point loop cannot finish: status 1, not silence."""
    b = synth.make_batch(3, Bsz=256)
    try:
        solver.set_options(warm_start=0, presolve=0, polish=0, max_iter=1)
        g = solver.solve_batch(*args(b), b["x0"])
        assert np.all(g["status"] == 1) and np.all(g["iters"] == 1)
        solver.set_options(kernel=KERNEL_GENERIC)
        g = solver.solve_batch(*args(b), b["x0"])
        assert np.all(g["status"] == 1)
        with pytest.raises(Exception):
            solver.set_options(max_iter=0)
        with pytest.raises(Exception):
            solver.set_options(eps=2.0)
    finally:
        solver.set_options(kernel=KERNEL_AUTO, warm_start=-1, presolve=-1, polish=1, max_iter=50, eps=1e-12)


# ---------------- the one-instance-per-workgroup kernel (32 < n <= 128) ----------------
def test_workgroup_kernel_dispatch_and_agreement(solver):
    """C5 goes to the workgroup kernel by default; forced on C4's dims it agrees with the register-resident
    specialisation and with the generic kernel; out of its range it refuses instead of falling back."""
    from lq_mpc_amd._lib import KERNEL_WORKGROUP
    b5, b4, b3 = synth.make_batch(5, Bsz=48), synth.make_batch(4, Bsz=96), synth.make_batch(3, Bsz=8)
    try:
        g5 = solver.rollout_batch(5, *args(b5), b5["x0"], b5["A_true"], b5["B_true"], want_traj=True)
        assert solver.last_kernel() == "lqmpc_wg_kernel"
        solver.set_options(kernel=KERNEL_GENERIC)
        r5 = solver.rollout_batch(5, *args(b5), b5["x0"], b5["A_true"], b5["B_true"], want_traj=True)
        assert solver.last_kernel() == "lqmpc_generic_kernel"
        assert rel(g5["J_T"], r5["J_T"]) < TIGHT and u_err(g5["U"], r5["U"]) < RTOL
        solver.set_options(kernel=KERNEL_AUTO)
        s4 = solver.solve_batch(*args(b4), b4["x0"])
        assert "spec" in solver.last_kernel() or "r64" in solver.last_kernel()
        solver.set_options(kernel=KERNEL_WORKGROUP)
        w4 = solver.solve_batch(*args(b4), b4["x0"])
        assert solver.last_kernel() == "lqmpc_wg_kernel"
        assert rel(w4["V_N"], s4["V_N"]) < TIGHT and u_err(w4["u_0"], s4["u_0"]) < RTOL
        with pytest.raises(Exception, match="workgroup"):
            solver.solve_batch(*args(b3), b3["x0"])             # n = 20: below the kernel's range
    finally:
        solver.set_options(kernel=KERNEL_AUTO)


@pytest.mark.parametrize("mix", ["default", "hard"])
def test_workgroup_kernel_interior_chunks_agree_with_the_stepwise_loop(solver, mix, golden_dir):
    """C5 rollouts without trajectories take the barrier-free chunks of interior steps (interior_steps<8,4>: state per wavefront,
    first exit found by one reduction, wave 0 repeats a partial chunk); with trajectories the one-barrier step loop runs.  Same
    closed loop, same order of additions: the costs agree to rounding, and both agree with the oracle."""
    b = synth.make_batch(5, Bsz=96, fixture_dir=golden_dir, mix=mix)
    T = 30 if mix == "default" else 12
    chunked = solver.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"])
    assert solver.last_kernel() == "lqmpc_wg_kernel"
    stepwise = solver.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"], want_traj=True)
    ref = orc.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"])
    assert np.all(chunked["status"] == 0) and np.all(stepwise["status"] == 0)
    assert rel(chunked["J_T"], stepwise["J_T"]) < 1e-12
    assert np.array_equal(chunked["iters"], stepwise["iters"])
    assert rel(chunked["J_T"], ref["J_T"]) < TIGHT


def test_16_lane_row_layout_whole_batch_tiered_and_hand_back(solver):
    """Rollouts of the shapes it serves run entirely in the 16-lane-row layout (lqmpc_r16_body.h; both of its builds,
    options.r16_build = 1/0); the packed family's tiered kernel uses it for its hardest instances (options.layout = 0 + nwide).  References, an off-centre box and
    per-instance plants go through it; with its iteration cap forced to 1 it hands the constrained instances back
    (status 3 internally) and the packed kernel's second pass must restore every one."""
    rng = np.random.default_rng(7)
    nx, nu, N, Bsz, T = 4, 2, 10, 2048, 12
    b = synth.make_batch(3, Bsz=Bsz)
    lb, ub = np.array([-0.08, -0.1]), np.array([0.1, 0.05])
    xr, ur = 0.05 * rng.standard_normal((nx, N)), 0.02 * rng.standard_normal((nu, N))
    At = np.ascontiguousarray(b["A"] + 0.01 * rng.standard_normal(b["A"].shape))
    Bt = np.ascontiguousarray(b["B"] + 0.01 * rng.standard_normal(b["B"].shape))
    a = (N, b["A"], b["B"], b["Q"], b["R"], 3.0 * b["P"], lb, ub)
    ref = orc.rollout_batch(T, *a, b["x0"], At, Bt, xr, ur, want_traj=True)
    try:
        solver.set_options(order=1)
        for layout, cap, nwide, lat, name in ((-1, 12, -1, 1, "r16"), (-1, 1, -1, 1, "r16"), (-1, 12, -1, 0, "r16"),
                                              (-1, 1, -1, 0, "r16"), (0, 12, 1024, -1, "tiered"),
                                              (0, 1, 1024, -1, "tiered"), (0, 12, 2048, -1, "tiered")):
            solver.set_options(layout=layout, r16_maxit=cap, nwide=nwide, r16_build=lat)
            g = solver.rollout_batch(T, *a, b["x0"], At, Bt, xr, ur, want_traj=True)
            assert name in solver.last_kernel()
            assert np.all(g["status"] == 0)
            assert rel(g["J_T"], ref["J_T"]) < TIGHT and u_err(g["U"], ref["U"]) < RTOL and np.abs(g["X"] - ref["X"]).max() < 1e-7
    finally:
        solver.set_options(order=-1, layout=-1, r16_maxit=12, nwide=-1, r16_build=-1)


@pytest.mark.parametrize("bsz", [96, 5000])
def test_one_instance_per_wavefront_hand_back_walks_the_list(solver, bsz, golden_dir):
    """C4's mapping (one instance per wavefront): with the iteration cap of the 16-lane-row family forced to 1 the constrained
    instances are handed back, and the packed kernel's pass over the device-side list -- a bounded grid whose workgroups walk the
    list (lqmpc_spec_list_kernel), 5000 instances here against its 2048 workgroups -- must restore every one; one-shot solves too."""
    b = synth.make_batch(4, Bsz=bsz, fixture_dir=golden_dir, mix="hard")
    T = 6
    sub = np.arange(min(bsz, 256))
    bs = dict(b, A=np.ascontiguousarray(b["A"][:, :, sub]), B=np.ascontiguousarray(b["B"][:, :, sub]))
    ref = orc.rollout_batch(T, *args(bs), np.ascontiguousarray(b["x0"][:, sub]), b["A_true"], b["B_true"])
    ref1 = orc.solve_batch(*args(bs), np.ascontiguousarray(b["x0"][:, sub]))
    try:
        full = solver.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"])
        assert "r64" in solver.last_kernel()
        solver.set_options(r16_maxit=1)
        g = solver.rollout_batch(T, *args(b), b["x0"], b["A_true"], b["B_true"])
        g1 = solver.solve_batch(*args(b), b["x0"])
    finally:
        solver.set_options(r16_maxit=12)
    assert np.all(g["status"] == 0) and np.all(g1["status"] == 0)
    assert rel(g["J_T"][sub], ref["J_T"]) < TIGHT and rel(g["J_T"], full["J_T"]) < 1e-9
    assert rel(g1["V_N"][sub], ref1["V_N"]) < TIGHT and u_err(g1["u_0"][:, sub], ref1["u_0"]) < RTOL


def test_sweep_batch_is_max_vn_plus_rollout(solver):
    """lqmpc_sweep_batch = lqmpc_max_vn_batch + lqmpc_rollout_batch for the same models: fused in one launch on the
    16-lane-row layout (C3 and the reference's shapes), two launches elsewhere (C4 shape here), also after a forced hand-back."""
    for cfg, Bsz, fused in ((3, 1500, True), (2, 1000, True), (4, 96, False), (5, 24, False)):
        b = synth.make_batch(cfg, Bsz=Bsz)
        a = (b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["lb"], b["ub"])
        x0s = np.ascontiguousarray(1.5 * b["x0"][:, :6])
        mv = orc.max_vn_batch(*a, x0s)
        jt = orc.rollout_batch(12, *a, b["x0"], b["A_true"], b["B_true"])["J_T"]
        for cap, lat in (((12, 1), (1, 1), (12, 0), (1, 0)) if fused else ((12, -1),)):
            solver.set_options(r16_maxit=cap, r16_build=lat)
            g = solver.sweep_batch(12, *a, b["x0"], x0s, b["A_true"], b["B_true"])
            assert ("r16" in solver.last_kernel()) == fused
            assert np.all(g["status"] == 0)
            assert rel(g["M_V"], mv) < TIGHT and rel(g["J_T"], jt) < TIGHT
            if cap == 12 and lat != 0: g0 = g
        solver.set_options(r16_maxit=12, r16_build=-1)
        m2 = solver.max_vn_batch(*a, x0s)
        r2 = solver.rollout_batch(12, *a, b["x0"], b["A_true"], b["B_true"])
        assert rel(g0["M_V"], m2["M_V"]) < 1e-12 and rel(g0["J_T"], r2["J_T"]) < 1e-12
        assert np.array_equal(g0["iters"], m2["iters"] + r2["iters"])        # iters = the sum of the two parts


def test_sweep_hand_back_merges_status_and_iters(solver):
    """Fused sweep with every constrained QP handed back (r16_maxit = 0) and a one-iteration interior-point budget on the packed
    kernel's second pass: status must be the worse of the max-V_N and rollout parts and iters their sum, exactly as
    lqmpc_max_vn_batch + lqmpc_rollout_batch report them (a failed max-V_N must not be overwritten by the rollout's status)."""
    b = synth.make_batch(3, Bsz=1500)
    a = (b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["lb"], b["ub"])
    x0s = np.ascontiguousarray(6.0 * b["x0"][:, :6])                       # far outside: the open-loop QPs are constrained
    x0 = np.ascontiguousarray(0.05 * b["x0"])                              # the rollouts mostly are not
    try:
        solver.set_options(r16_maxit=0, max_iter=1, polish=0)
        g = solver.sweep_batch(8, *a, x0, x0s, b["A_true"], b["B_true"])
        assert "r16" in solver.last_kernel()
        m = solver.max_vn_batch(*a, x0s)
        r = solver.rollout_batch(8, *a, x0, b["A_true"], b["B_true"])
        assert np.any(m["status"] == 1) and np.any(r["status"] == 0)       # the case the merge exists for
        assert np.array_equal(g["status"], np.maximum(m["status"], r["status"]))
        assert np.array_equal(g["iters"], m["iters"] + r["iters"])
    finally:
        solver.set_options(r16_maxit=12, max_iter=50, polish=1)


def test_c4_eight_way_sharded_form_equals_the_single_batch(solver, golden_dir):
    """BASELINE config 4 is 262 144 systems cut over 8 GPUs (32 768 per rank, lq_mpc_amd.dist.shard_batch).  The eight shards,
    rolled out one after the other on the one GPU here, must give the single-batch result bit for bit (no result depends on
    which instances share a launch or on their position in the difficulty order) -- the property the 8-GPU run relies on."""
    from lq_mpc_amd import dist as ld
    b = synth.make_batch(4, fixture_dir=golden_dir)
    assert b["Bsz"] == 262144
    full = solver.rollout_batch(30, *args(b), b["x0"], b["A_true"], b["B_true"])
    assert np.all(full["status"] == 0)
    parts = []
    for r in range(8):
        sh = ld.shard_batch(b, r, 8)
        assert sh["A"].shape[-1] == 32768 and sh["shard"] == (r * 32768, (r + 1) * 32768)
        g = solver.rollout_batch(30, *args(sh), sh["x0"], sh["A_true"], sh["B_true"])
        assert "r64" in solver.last_kernel() and np.all(g["status"] == 0)
        parts.append(g["J_T"])
    assert np.array_equal(np.concatenate(parts), full["J_T"])
    idx = np.random.default_rng(3).choice(b["Bsz"], 256, replace=False)
    sub = dict(b, A=np.ascontiguousarray(b["A"][:, :, idx]), B=np.ascontiguousarray(b["B"][:, :, idx]))
    ref = orc.rollout_batch(30, *args(sub), np.ascontiguousarray(b["x0"][:, idx]), b["A_true"], b["B_true"])
    assert rel(full["J_T"][idx], ref["J_T"]) < TIGHT


def test_ordered_rollout_growing_batch_on_one_handle(golden_dir):
    """A handle that has run an ordered (probe + bucket order) rollout re-allocates its counter / hand-back buffer when the batch
    grows: the no-fill path of the order counters must notice (by capacity, not by address) and refill, or the scatter writes out
    of bounds.  8 192 then 65 536 instances on the SAME fresh handle, each against a separate handle's natural-order run."""
    s = BatchSolver(0)
    s2 = BatchSolver(0, order=0)
    try:
        for bsz in (8192, 65536, 8192):
            b = synth.make_batch(3, Bsz=bsz, fixture_dir=golden_dir)
            a = (*args(b), b["x0"], b["A_true"], b["B_true"])
            g = s.rollout_batch(12, *a)
            r = s2.rollout_batch(12, *a)
            assert np.all(g["status"] == 0) and np.all(r["status"] == 0)
            assert rel(g["J_T"], r["J_T"]) < 1e-12
        m = 512
        b = synth.make_batch(3, Bsz=65536, fixture_dir=golden_dir)
        ref = orc.rollout_batch(12, b["N"], b["A"][:, :, :m].copy(), b["B"][:, :, :m].copy(), b["Q"], b["R"], b["P"], b["lb"], b["ub"],
                                b["x0"][:, :m].copy(), b["A_true"], b["B_true"])
        g = s.rollout_batch(12, *args(b), b["x0"], b["A_true"], b["B_true"])
        assert rel(g["J_T"][:m], ref["J_T"]) < TIGHT
    finally:
        s.close(); s2.close()


@pytest.mark.parametrize("nx,nu,N", [(3, 2, 6), (1, 1, 1), (2, 1, 12), (4, 2, 12), (4, 2, 24), (5, 3, 4), (8, 4, 8), (6, 3, 15), (3, 3, 5)])
def test_run_time_compiled_shapes(solver, nx, nu, N):
    """Shapes without a prebuilt instantiation get the 16-lane-row kernel compiled at run time (lqmpc_jit.hip; utils_class.py:23, 62:
    the reference takes any N): every entry point, the ordered walk (probe compiled too), the hand-back to the generic kernel over
    the device-side list (iteration cap forced to 1), and options.jit = 0 (generic kernel) must all agree with the oracle."""
    rng = np.random.default_rng(1000 * nx + 10 * N + nu)
    Bsz, T, m = 8192 + 5, 8, 320
    A = rng.standard_normal((nx, nx, Bsz))
    A *= rng.uniform(0.4, 1.15, Bsz) / np.abs(np.linalg.eigvals(A.transpose(2, 0, 1))).max(axis=1)
    B = rng.standard_normal((nx, nu, Bsz)) * rng.uniform(0.2, 1.5, (1, 1, Bsz))
    M = rng.standard_normal((nx, nx)); Q = M @ M.T / nx + np.eye(nx)
    M = rng.standard_normal((nu, nu)); R = 0.3 * (M @ M.T / nu + np.eye(nu))
    P = 2.5 * Q
    lb, ub = -rng.uniform(0.05, 0.3, nu), rng.uniform(0.05, 0.3, nu)
    x0 = rng.standard_normal((nx, Bsz)) * rng.choice([1e-2, 0.3, 1.0, 3.0], Bsz)
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    At = np.ascontiguousarray(0.95 * A); Bt = B
    x0s = np.ascontiguousarray(rng.standard_normal((nx, 5)))
    umax = float(np.max(np.maximum(-lb, ub)))
    sub = lambda a: np.ascontiguousarray(a[..., :m])
    r1 = orc.solve_batch(N, sub(A), sub(B), Q, R, P, lb, ub, sub(x0))
    r2 = orc.rollout_batch(T, N, sub(A), sub(B), Q, R, P, lb, ub, sub(x0), sub(At), sub(Bt), want_traj=True)
    r3 = orc.max_vn_batch(N, sub(A), sub(B), Q, R, P, lb, ub, x0s)
    ok = np.isfinite(r2["J_T"]) & (np.abs(r2["X"]).max(axis=(0, 1)) < 1e6)
    assert ok.mean() > 0.5

    def check(g1, g2, g3, g4):
        assert np.all(g1["status"][:m] == 0) and np.all(g2["status"][:m][ok] == 0)
        assert rel(g1["V_N"][:m], r1["V_N"]) < 1e-7 and u_err(g1["u_0"][:, :m], r1["u_0"], umax) < RTOL
        assert rel(g2["J_T"][:m][ok], r2["J_T"][ok]) < 1e-6
        assert rel(g3["M_V"][:m], r3) < 1e-7
        assert rel(g4["M_V"][:m], r3) < 1e-7 and rel(g4["J_T"][:m][ok], r2["J_T"][ok]) < 1e-6
    a = (N, A, B, Q, R, P, lb, ub)
    try:
        for cap in (12, 1):                       # 1: every constrained QP is handed back -> lqmpc_generic_list_kernel
            solver.set_options(r16_maxit=cap)
            g1 = solver.solve_batch(*a, x0); k1 = solver.last_kernel()
            g2 = solver.rollout_batch(T, *a, x0, At, Bt); k2 = solver.last_kernel()           # 8197 instances, T = 8: the ordered walk
            g3 = solver.max_vn_batch(*a, x0s); k3 = solver.last_kernel()
            g4 = solver.sweep_batch(T, *a, x0, x0s, At, Bt); k4 = solver.last_kernel()
            for k in (k1, k2, k3, k4):
                assert "jit" in k and (f"<{nx},{nu},{N}>" in k) and ("r16" in k) == (N * nu <= 32), k
            check(g1, g2, g3, g4)
        solver.set_options(r16_maxit=12, jit=0)
        h1 = solver.solve_batch(N, sub(A), sub(B), Q, R, P, lb, ub, sub(x0))
        assert "jit" not in solver.last_kernel()
        assert rel(h1["V_N"], g1["V_N"][:m]) < 1e-7
    finally:
        solver.set_options(r16_maxit=12, jit=-1)
