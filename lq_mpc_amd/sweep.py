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

    @staticmethod
    def _check_bounds_status(r):
        if np.any(r["status"] != 0):
            cnt = {int(k): int(np.sum(r["status"] == k)) for k in np.unique(r["status"]) if k != 0}
            raise RuntimeError(f"lqmpc_bounds_batch: models per non-zero status {cnt} (1 = dlqr doubling not settled, 2 = not "
                               "stabilisable, 3 = the bound formulas leave the reals: rho(A - BK) + 0.4 > 1, utils.py:358)")

    def _pool(self, k):
        """k handles (own stream each) for the concurrent passes of data_generation: the reference's six passes (error levels, five
        horizons) are independent, and each is latency-bound on its own (a few hundred instances)."""
        if not hasattr(self, "_handles"):
            self._handles = []
        from .mpc import BatchSolver
        while len(self._handles) < k:
            self._handles.append(BatchSolver(self._s().device))
        return self._handles[:k]

    def data_generation(self, N_points, ext_radius_max, info_ref, p=None, save_path=None, concurrent=True):
        """utils_class.py:766-959.  concurrent: the error-level pass and the five horizon passes (one fused sweep launch + one
        bounds launch each) run on handles / streams of their own from a thread pool (ctypes releases the GIL); False: one after
        the other on the solver's own handle."""
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
        # horizon sweep on error column index_sys = 4 (utils_class.py:880-883), zero references (887-888)
        A4 = np.ascontiguousarray(self.A_true[:, :, None] + self.error_A[:, :, :, 4])
        B4 = np.ascontiguousarray(self.B_true[:, :, None] + self.error_B[:, :, :, 4])
        x04 = np.repeat(x_start[:, None], self.N_sys, axis=1)
        lev = np.tile(self.error_vec, self.N_sys)             # the error level of instance (j, i) is error_vec[i]
        lev4 = np.full(self.N_sys, self.e_nominal)

        def one_pass(sv, N, Am, Bm, x0m, xr, ur, levels):
            """M_V (8 open-loop solves per system) and the closed-loop cost in one launch (lqmpc_sweep_batch), then the bound
            coefficients of the same models (lqmpc_bounds_batch)"""
            res = sv.sweep_batch(self.N_mpc, N, Am, Bm, Q, R, Q, lb, ub, x0m, x0_vec, self.A_true, self.B_true, xr, ur)
            co = None
            if p is not None:
                r = sv.bounds_batch(N, Am, Bm, Q, R, lb, ub, levels, levels, res["M_V"], x_start, p, V_expert)
                self._check_bounds_status(r)
                co = (r["alpha"], r["beta"], r["xi"], r["bound"])
            return res["M_V"], res["J_T"], co
        jobs = [(self.N_nominal, A, B, x0, info_ref.get("x_ref"), info_ref.get("u_ref"), lev)]
        jobs += [(int(Nh), A4, B4, x04, None, None, lev4) for Nh in self.horizon]
        if concurrent:
            from concurrent.futures import ThreadPoolExecutor
            handles = self._pool(len(jobs))
            if not hasattr(self, "_executor"):
                self._executor = ThreadPoolExecutor(max_workers=len(jobs))
            results = list(self._executor.map(lambda hj: one_pass(hj[0], *hj[1]), zip(handles, jobs)))
        else:
            results = [one_pass(s, *j) for j in jobs]
        M_V_error, J, co_err = results[0]
        true_cost_error = J.reshape(self.N_sys, n_err)
        M_V_error = M_V_error.reshape(self.N_sys, n_err)
        true_cost_horizon = np.stack([r[1] for r in results[1:]], axis=1)
        M_V_horizon = np.stack([r[0] for r in results[1:]], axis=1)
        out = {"error": self.error_vec, "horizon": self.horizon, "V_expert": V_expert,
               "true_cost_error": true_cost_error, "true_cost_horizon": true_cost_horizon}
        if p is not None:
            for k, v in zip(("alpha", "beta", "xi", "bound"), co_err):
                out[f"{k}_table_error"] = v.reshape(self.N_sys, n_err)
            for j, k in enumerate(("alpha", "beta", "xi", "bound")):
                out[f"{k}_table_horizon"] = np.stack([r[2][j] for r in results[1:]], axis=1)
            if save_path is not None:
                np.savez(save_path, **out)                                             # utils_class.py:958
        out.update({"M_V_error": M_V_error, "M_V_horizon": M_V_horizon, "x0_vec": x0_vec})
        return out
