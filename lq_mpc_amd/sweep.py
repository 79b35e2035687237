"""Batched counterpart of the reference's multi-system sweep (the batch axis of the hot path).

Mirrors LQ_RDP_Behavior_Multiple (/root/reference/utils_class.py:685-959) for the part of
data_generation that is the MPC hot path:
    V_expert            utils_class.py:786
    M_V (max of 8 V_N)  utils_class.py:813-824, 896-907
    true_cost_error     utils_class.py:828-833
    true_cost_horizon   utils_class.py:911-916
The reference issues 57 001 sequential cvxpy solves for this; here every (error level, system) pair
is one instance of ONE batched launch per horizon, read straight from the reference's .npy files
(their memory layout already is the library's instance-minor layout, utils_class.py:749-750).
The bound coefficients alpha / beta / xi / bound of every system (control.dlqr + energy_decreasing + energy_bound,
utils_class.py:837-859, 920-942) come from the GPU as well (lqmpc_bounds_batch, one instance per lane); with them
data_generation returns, and optionally saves, the same 13 arrays as the reference's data_lq_mpc_multipleSys.npz
(utils_class.py:944-958).
"""
import math
import os

import numpy as np

from .mpc import box_from_Fu, default_solver


def circle_generator(N_points, ratio_ext_radius, my_base, Q, seed=0):
    """N_points initial states on the level set x'Qx = (ratio*sqrt(base))^2   (utils.py:683-704).

    n_x = 2 is the reference's construction, value for value: the point (r, 0) rotated by theta_k = linspace(0, 2(1-1/N)pi, N)
    and mapped through the inverse of the upper Cholesky factor of Q.  The reference is hard-wired to two states; for n_x > 2
     the same circle is drawn in floor(n_x/2) planes of a seeded random orthonormal basis, point k in plane
    k mod floor(n_x/2), so every point still satisfies x'Qx = r^2 and the set is reproducible from `seed`."""
    Q = np.asarray(Q, dtype=np.float64)
    nx = Q.shape[0]
    root_Q = np.linalg.cholesky(Q).T            # scipy cho_factor's default (upper) factor
    r = ratio_ext_radius * math.sqrt(my_base)
    theta = np.linspace(0, 2 * (1 - 1 / N_points) * math.pi, N_points)
    if nx == 2:
        basis = np.eye(2)
    else:
        q, rr = np.linalg.qr(np.random.default_rng(seed).standard_normal((nx, nx)))
        basis = q * np.sign(np.diag(rr))
    planes = max(nx // 2, 1)
    pts = np.zeros((nx, N_points))
    for k, t in enumerate(theta):
        p = k % planes
        e1 = basis[:, 2 * p]
        e2 = basis[:, 2 * p + 1] if nx > 1 else 0.0
        pts[:, k] = r * (math.cos(t) * e1 + math.sin(t) * e2)
    return np.linalg.solve(root_Q, pts)


class LQ_RDP_Behavior_Multiple:
    """Same constructor arguments as utils_class.py:691-692 (plus data_dir / solver)."""

    def __init__(self, info_opc, info_N, info_e_pow, N_sys, norm_type, errM_import=True, data_dir=".", solver=None):
        self.A_true = np.asarray(info_opc["A"], dtype=np.float64)
        self.B_true = np.asarray(info_opc["B"], dtype=np.float64)
        self.Q = np.asarray(info_opc["Q"], dtype=np.float64)
        self.R = np.asarray(info_opc["R"], dtype=np.float64)
        self.F_u = np.asarray(info_opc["F_u"], dtype=np.float64)
        self.lb, self.ub = box_from_Fu(self.F_u)
        self.N_sys = 5 * N_sys                                   # utils_class.py:726
        self.N_min, self.N_max = info_N["N_min"], info_N["N_max"]
        self.N_nominal, self.N_opc, self.N_mpc = info_N["N_nominal"], info_N["N_opc"], info_N["N_mpc"]
        self.e_min, self.e_max, self.e_nominal = info_e_pow["e_min"], info_e_pow["e_max"], info_e_pow["e_nominal"]
        self.horizon = np.arange(self.N_min, self.N_max + 1)
        self.error_vec = np.linspace(self.e_min, self.e_max, 10)
        if not errM_import:
            raise NotImplementedError("the reference's unseeded perturbation generator (utils.py:760-847) is out of scope; "
                                      "pass the shipped error_{A,B}_<norm>.npy files")
        self.error_A = np.load(os.path.join(data_dir, f"error_A_{norm_type}.npy"))      # (nx, nx, N_sys, 10)
        self.error_B = np.load(os.path.join(data_dir, f"error_B_{norm_type}.npy"))      # (nx, nu, N_sys, 10)
        self._solver = solver
        self._eps_lqr = None

    def _s(self):
        return self._solver if self._solver is not None else default_solver()

    @property
    def epsilon_lqr(self):
        """local_radius(F_u, -K_lqr, Q) of the true system (utils_class.py:761-764): dlqr and the radius from the GPU."""
        if self._eps_lqr is None:
            nx = self.A_true.shape[0]
            r = self._s().bounds_batch(1, self.A_true[:, :, None], self.B_true[:, :, None], self.Q, self.R, self.lb, self.ub,
                                       0.0, 0.0, None, np.zeros(nx), np.ones(3), 0.0)
            # status 3 = the bound formulas leave the reals (rho(A - BK) + 0.4 > 1, utils.py:358) -- K and eps are valid there, and
            # the reference computes local_radius for any stabilisable system (utils_class.py:761-764)
            if r["status"][0] not in (0, 3):
                raise RuntimeError(f"dlqr of the true system failed (lqmpc_bounds_batch status {int(r['status'][0])}: "
                                   "1 = doubling iteration not settled, 2 = not stabilisable)")
            self._eps_lqr, self.K_lqr = float(r["eps"][0]), r["K"][:, :, 0].copy()
        return self._eps_lqr

    def _bound_tables(self, N, A_stack, B_stack, e_level, M_V, x_start, V_expert, p):
        """alpha, beta, xi, bound for a batch of models sharing horizon N (utils_class.py:837-859): one launch."""
        r = self._s().bounds_batch(N, A_stack, B_stack, self.Q, self.R, self.lb, self.ub, e_level, e_level, M_V, x_start, p, V_expert)
        if np.any(r["status"] != 0):
            cnt = {int(k): int(np.sum(r["status"] == k)) for k in np.unique(r["status"]) if k != 0}
            raise RuntimeError(f"lqmpc_bounds_batch: models per non-zero status {cnt} (1 = dlqr doubling not settled, 2 = not "
                               "stabilisable, 3 = the bound formulas leave the reals: rho(A - BK) + 0.4 > 1, utils.py:358)")
        return r["alpha"], r["beta"], r["xi"], r["bound"]

    def data_generation(self, N_points, ext_radius_max, info_ref, p=None, save_path=None):
        s = self._s()
        nx, nu = self.B_true.shape
        Q, R, lb, ub = self.Q, self.R, self.lb, self.ub
        x0_vec = circle_generator(N_points, ext_radius_max, self.epsilon_lqr, Q)         # utils_class.py:782
        x_start = x0_vec[:, 1]                                                            # utils_class.py:783
        V_expert = float(s.solve_batch(self.N_opc, self.A_true[:, :, None], self.B_true[:, :, None], Q, R, Q, lb, ub,
                                       x_start[:, None], info_ref.get("x_ref_long"), info_ref.get("u_ref_long"))["V_N"][0])
        n_err = len(self.error_vec)
        # error sweep: every (system j, level i) pair is one instance; the file layout is (.., j, i)
        A = np.ascontiguousarray((self.A_true[:, :, None, None] + self.error_A).reshape(nx, nx, -1))
        B = np.ascontiguousarray((self.B_true[:, :, None, None] + self.error_B).reshape(nx, nu, -1))
        Bsz = A.shape[2]
        x0 = np.repeat(x_start[:, None], Bsz, axis=1)
        N = self.N_nominal
        # M_V (8 open-loop solves per system) and the closed-loop cost share one launch: lqmpc_sweep_batch
        res = s.sweep_batch(self.N_mpc, N, A, B, Q, R, Q, lb, ub, x0, x0_vec, self.A_true, self.B_true,
                            info_ref.get("x_ref"), info_ref.get("u_ref"))
        M_V_error, J = res["M_V"], res["J_T"]
        true_cost_error = J.reshape(self.N_sys, n_err)
        M_V_error = M_V_error.reshape(self.N_sys, n_err)
        # horizon sweep on error column index_sys = 4 (utils_class.py:880-883), zero references (887-888)
        A4 = np.ascontiguousarray(self.A_true[:, :, None] + self.error_A[:, :, :, 4])
        B4 = np.ascontiguousarray(self.B_true[:, :, None] + self.error_B[:, :, :, 4])
        x04 = np.repeat(x_start[:, None], self.N_sys, axis=1)
        true_cost_horizon = np.zeros((self.N_sys, len(self.horizon)))
        M_V_horizon = np.zeros((self.N_sys, len(self.horizon)))
        for i, Nh in enumerate(self.horizon):
            Nh = int(Nh)
            res = s.sweep_batch(self.N_mpc, Nh, A4, B4, Q, R, Q, lb, ub, x04, x0_vec, self.A_true, self.B_true)
            M_V_horizon[:, i], true_cost_horizon[:, i] = res["M_V"], res["J_T"]
        out = {"error": self.error_vec, "horizon": self.horizon, "V_expert": V_expert,
               "true_cost_error": true_cost_error, "true_cost_horizon": true_cost_horizon}
        if p is not None:
            # the scalar bound coefficients (host side); the error level of instance (j, i) is error_vec[i]
            lev = np.tile(self.error_vec, self.N_sys)
            al, be, xi, bd = self._bound_tables(N, A, B, lev, M_V_error.reshape(-1), x_start, V_expert, p)
            for k, v in (("alpha", al), ("beta", be), ("xi", xi), ("bound", bd)):
                out[f"{k}_table_error"] = v.reshape(self.N_sys, n_err)
            for k in ("alpha", "beta", "xi", "bound"):
                out[f"{k}_table_horizon"] = np.zeros((self.N_sys, len(self.horizon)))
            lev4 = np.full(self.N_sys, self.e_nominal)
            for i, Nh in enumerate(self.horizon):
                al, be, xi, bd = self._bound_tables(int(Nh), A4, B4, lev4, M_V_horizon[:, i], x_start, V_expert, p)
                out["alpha_table_horizon"][:, i], out["beta_table_horizon"][:, i] = al, be
                out["xi_table_horizon"][:, i], out["bound_table_horizon"][:, i] = xi, bd
            if save_path is not None:
                np.savez(save_path, **out)                                             # utils_class.py:958
        out.update({"M_V_error": M_V_error, "M_V_horizon": M_V_horizon, "x0_vec": x0_vec})
        return out
