"""CPU: the host oracle of the bound coefficients (oracle/bounds_np.py) against the reference's npz.

M_V and V_expert come from the QP oracle here (CPU); tests/test_gpu_bounds.py checks the HIP kernel (lqmpc_bounds_batch) against
this oracle and against the same npz."""
import os

import numpy as np
import pytest

from oracle import bounds_np as bounds
from oracle import oracle as orc

A0 = np.array([[1.0, 0.7], [0.12, 0.4]])
B0 = np.array([[1.0], [1.2]])
Q = 2.0 * np.eye(2)
R = np.eye(1)
F_U = np.vstack((10 * np.eye(1), -10 * np.eye(1)))
P_PAIR = np.array([0.1, 1, 0.6])


def test_box_bar_u_closed_forms():
    assert bounds.box_bar_u([-0.1], [0.1]) == pytest.approx((0.01, 0.04))
    assert bounds.box_bar_u([-0.2, -0.1], [0.1, 0.3]) == pytest.approx((0.04 + 0.09, 0.09 + 0.16))


def test_tables_against_reference_npz(golden_dir):
    d = np.load(os.path.join(golden_dir, "data_lq_mpc_multipleSys.npz"))
    eA = np.load(os.path.join(golden_dir, "error_A_f.npy"))
    eB = np.load(os.path.join(golden_dir, "error_B_f.npy"))
    lb, ub = np.array([-0.1]), np.array([0.1])
    K = orc.dlqr_gain(A0, B0, Q, R)[0]
    eps = orc.local_radius(F_U, -K, Q)
    x0_vec = orc.circle_generator(8, 1.5, eps, Q)
    xs = x0_vec[:, 1]
    V_expert = orc.solve(30, A0, B0, Q, R, Q, lb, ub, xs)["V_N"]
    A = (A0[:, :, None, None] + eA).reshape(2, 2, 1000)
    B = (B0[:, :, None, None] + eB).reshape(2, 1, 1000)
    MV = orc.max_vn_batch(7, A, B, Q, R, Q, lb, ub, x0_vec).reshape(100, 10)
    err = d["error"]
    xi = np.zeros((100, 10)); al = np.zeros((100, 10)); be = np.zeros((100, 10)); bd = np.zeros((100, 10))
    for i in range(10):
        for j in range(0, 100, 7):          # a sample of rows keeps the CPU suite fast
            Am, Bm = A0 + eA[:, :, j, i], B0 + eB[:, :, j, i]
            Km = orc.dlqr_gain(Am, Bm, Q, R)[0]
            ed = bounds.energy_decreasing(7, Am, Bm, Q, R, F_U, err[i], err[i], -Km, MV[j, i])
            eb = bounds.energy_bound(7, Am, Bm, Q, R, lb, ub, err[i], err[i], xs, P_PAIR)
            xi[j, i], al[j, i], be[j, i] = ed["xi"], eb["alpha"], eb["beta"]
            bd[j, i] = (eb["alpha"] * V_expert + eb["beta"]) / (1 - ed["xi"] - ed["eta"])
    rows = np.arange(0, 100, 7)
    np.testing.assert_allclose(xi[rows], d["xi_table_error"][rows], rtol=1e-10)
    np.testing.assert_allclose(al[rows], d["alpha_table_error"][rows], rtol=1e-11)
    np.testing.assert_allclose(be[rows], d["beta_table_error"][rows], rtol=1e-11)
    np.testing.assert_allclose(bd[rows], d["bound_table_error"][rows], rtol=1e-9)
