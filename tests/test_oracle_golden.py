"""CPU: pin the oracle against the reference's own committed data.

Inputs  error_A_f.npy / error_B_f.npy  (read by the reference at utils_class.py:749-750)
Outputs data_lq_mpc_multipleSys.npz    (written by the reference at utils_class.py:944-958)
The reference's constants are those of working_example_multiple.py:13-58, 70-76.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

A0 = np.array([[1.0, 0.7], [0.12, 0.4]])
B0 = np.array([[1.0], [1.2]])
Q = 2.0 * np.eye(2)
R = np.eye(1)
F_U = np.vstack((10 * np.eye(1), -10 * np.eye(1)))


@pytest.fixture(scope="module")
def ref_data(golden_dir):
    d = np.load(os.path.join(golden_dir, "data_lq_mpc_multipleSys.npz"))
    eA = np.load(os.path.join(golden_dir, "error_A_f.npy"))
    eB = np.load(os.path.join(golden_dir, "error_B_f.npy"))
    return d, eA, eB


@pytest.fixture(scope="module")
def setup_consts():
    lb, ub = orc.box_from_Fu(F_U)
    K, _ = orc.dlqr_gain(A0, B0, Q, R)
    eps = orc.local_radius(F_U, -K, Q)
    x0_vec = orc.circle_generator(8, 1.5, eps, Q)
    return lb, ub, K, eps, x0_vec


def test_reference_constants(setup_consts):
    lb, ub, K, eps, x0_vec = setup_consts
    assert lb[0] == -0.1 and ub[0] == 0.1
    np.testing.assert_allclose(K, [[0.48363093, 0.45846723]], rtol=1e-7)
    assert abs(eps - 0.04503580745099056) < 1e-15
    np.testing.assert_allclose(x0_vec[:, 0], [0.22508950082659201, 0.0], rtol=1e-14, atol=1e-17)
    np.testing.assert_allclose(x0_vec[:, 1], [0.15916231240837822, 0.15916231240837819], rtol=1e-14)


def test_V_expert(ref_data, setup_consts):
    d, _, _ = ref_data
    lb, ub, _, _, x0_vec = setup_consts
    r = orc.solve(30, A0, B0, Q, R, Q, lb, ub, x0_vec[:, 1])
    # the reference's value comes from cvxpy's iterative back-end: agreement ~1e-12
    assert abs(r["V_N"] - float(d["V_expert"])) / float(d["V_expert"]) < 1e-11
    np.testing.assert_allclose(r["U"][0, :4], [-0.1, -0.06542449, -0.0048106, 0.00044005], atol=2e-8)


def test_true_cost_error_table(ref_data, setup_consts):
    d, eA, eB = ref_data
    lb, ub, _, _, x0_vec = setup_consts
    A = (A0[:, :, None, None] + eA).reshape(2, 2, 1000)
    B = (B0[:, :, None, None] + eB).reshape(2, 1, 1000)
    x0 = np.repeat(x0_vec[:, 1:2], 1000, 1)
    J = orc.rollout_batch(30, 7, A, B, Q, R, Q, lb, ub, x0, A0, B0)["J_T"].reshape(100, 10)
    np.testing.assert_allclose(J, d["true_cost_error"], rtol=1e-13)


@pytest.mark.parametrize("k,N", list(enumerate(range(6, 11))))
def test_true_cost_horizon_table(ref_data, setup_consts, k, N):
    d, eA, eB = ref_data
    lb, ub, _, _, x0_vec = setup_consts
    A = A0[:, :, None] + eA[:, :, :, 4]            # index_sys = 4, utils_class.py:880-883
    B = B0[:, :, None] + eB[:, :, :, 4]
    x0 = np.repeat(x0_vec[:, 1:2], 100, 1)
    J = orc.rollout_batch(30, N, A, B, Q, R, Q, lb, ub, x0, A0, B0)["J_T"]
    np.testing.assert_allclose(J, d["true_cost_horizon"][:, k], rtol=1e-13)


def test_norm2_files_do_not_match(ref_data, golden_dir, setup_consts):
    """The npz was generated with norm_type='f' (working_example_multiple.py:98-99): the _2 files differ."""
    d, _, _ = ref_data
    lb, ub, _, _, x0_vec = setup_consts
    eA = np.load(os.path.join(golden_dir, "error_A_2.npy"))
    eB = np.load(os.path.join(golden_dir, "error_B_2.npy"))
    A = (A0[:, :, None, None] + eA).reshape(2, 2, 1000)
    B = (B0[:, :, None, None] + eB).reshape(2, 1, 1000)
    x0 = np.repeat(x0_vec[:, 1:2], 1000, 1)
    J = orc.rollout_batch(30, 7, A, B, Q, R, Q, lb, ub, x0, A0, B0)["J_T"].reshape(100, 10)
    assert np.max(np.abs(J - d["true_cost_error"]) / d["true_cost_error"]) > 1e-6


def test_script_known_answers(setup_consts):
    """mpc_test.py:33-62 and working_example_single.py:62-66 run through the oracle (SURVEY 8(c) probe values)."""
    lb, ub, _, _, x0_vec = setup_consts
    x0 = np.array([0.1125, 0.19])
    r = orc.solve(20, A0, B0, Q, R, Q, lb, ub, x0)
    assert abs(r["V_N"] - 0.17375571642447996) < 1e-12 and abs(r["u_0"][0] + 0.1) < 1e-15
    s = orc.simulate(20, 6, A0, B0, Q, R, Q, lb, ub, x0, np.array([[1.01, 0.7], [0.12, 0.41]]), np.array([[1.0], [1.21]]))
    assert abs(s["J_T"] - 0.17571889572646185) < 1e-13
    mv = orc.max_vn_batch(6, A0[:, :, None], B0[:, :, None], Q, R, Q, lb, ub, x0_vec)
    assert abs(mv[0] - 0.2022946791688417) < 1e-13
