"""Host-side mirror of the reference's operator interface for the MPC hot path.

Same class names, positional signatures and return dictionaries as
/root/reference/utils_class.py:
    LQ_MPC_Controller(N, A, B, Q, R, P, F_u).solve(x0, x_ref, u_ref) -> {'u_0', 'V_N'}   (lines 18-91)
    LQ_MPC_Simulator(T, N, A, B, Q, R, P, F_u).simulate(x0, A_true, B_true, x_ref, u_ref)
        -> {'X', 'U', 'J_T'}                                                              (lines 213-285)
so callers written against the reference (LQ_RDP_Behavior*, mpc_test.py) run unchanged, plus the
batched entry points a GPU needs (the reference API is one instance per call):
    BatchSolver.solve_batch / rollout_batch / max_vn_batch  over instance-minor (SoA) arrays.
Everything executes in the HIP library behind include/lqmpc.h; there is no CPU path here.
"""
import ctypes

import numpy as np

from . import _lib


def box_from_Fu(F_u):
    """F_u u <= 1 with one non-zero per row (utils_class.py:81) -> (lb, ub).

    The reference accepts any polytope; every caller uses the box [10 I; -10 I].  Only box rows are
    supported here; anything else raises ValueError."""
    F_u = np.atleast_2d(np.asarray(F_u, dtype=np.float64))
    nu = F_u.shape[1]
    lb = np.full(nu, -np.inf)
    ub = np.full(nu, np.inf)
    for row in F_u:
        nz = np.flatnonzero(row)
        if nz.size != 1:
            raise ValueError("F_u must describe a box: exactly one non-zero per row")
        k = nz[0]
        if row[k] > 0:
            ub[k] = min(ub[k], 1.0 / row[k])
        else:
            lb[k] = max(lb[k], 1.0 / row[k])
    if not (np.all(np.isfinite(lb)) and np.all(np.isfinite(ub)) and np.all(lb < ub)):
        raise ValueError("F_u must bound every input from both sides (finite, non-empty box)")
    return lb, ub


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _ptr(a):
    """void* for the ABI: numpy array (host), int address (device), object with data_ptr() (torch)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return ctypes.c_void_p(a.data_ptr())
    return ctypes.c_void_p(int(a))


def _ref_or_none(r, rows, N):
    """References as the reference reads them: columns 0..N-1 of a (rows, >= N) array (utils_class.py:69, 75 index
    x_ref[:, i], u_ref[:, i] for i < N only, so wider arrays -- e.g. one long reference shared by several horizons -- are
    accepted and the tail is ignored).  All-zero references (every caller in the reference) are passed as NULL."""
    if r is None:
        return None
    r = np.asarray(r, dtype=np.float64)
    if r.ndim != 2 or r.shape[0] != rows or r.shape[1] < N:
        raise ValueError(f"expected a reference of shape ({rows}, >= {N}), got {r.shape}")
    r = np.ascontiguousarray(r[:, :N])
    return r if np.any(r) else None


class BatchSolver:
    """One handle = one GPU + one stream (include/lqmpc.h).  Not thread-safe."""

    def __init__(self, device=0, stream=None, **options):
        self._L = _lib.lib()
        if self._L.lqmpc_device_count() <= 0:
            raise _lib.LqmpcError("no MI355X/HIP device visible: the lq_mpc_amd product path is GPU-only")
        self._h = ctypes.c_void_p()
        if stream is None:
            _lib.check(self._L.lqmpc_create(int(device), ctypes.byref(self._h)))
        else:
            _lib.check(self._L.lqmpc_create_on_stream(int(device), ctypes.c_void_p(int(stream)), ctypes.byref(self._h)))
        self.device = int(device)
        if options:
            self.set_options(**options)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.lqmpc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- options ----
    def get_options(self):
        o = _lib.Options()
        _lib.check(self._L.lqmpc_get_options(self._h, ctypes.byref(o)))
        return {k: getattr(o, k) for k, _ in o._fields_}

    def set_options(self, **kw):
        o = _lib.Options()
        _lib.check(self._L.lqmpc_get_options(self._h, ctypes.byref(o)))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k!r}")
            setattr(o, k, v)
        _lib.check(self._L.lqmpc_set_options(self._h, ctypes.byref(o)))

    def sync(self):
        _lib.check(self._L.lqmpc_sync(self._h))

    def last_kernel(self):
        return self._L.lqmpc_last_kernel(self._h).decode()

    def reserve(self, nx, nu, N, Bsz, T=0):
        _lib.check(self._L.lqmpc_reserve(self._h, nx, nu, N, Bsz, T))

    def timer_begin(self):
        _lib.check(self._L.lqmpc_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_float()
        _lib.check(self._L.lqmpc_timer_end(self._h, ctypes.byref(ms)))
        return ms.value

    # ---- host-array entry points (numpy in, numpy out) ----
    @staticmethod
    def _dims(A, B):
        A = _f64(A)
        B = _f64(B)
        if A.ndim != 3 or B.ndim != 3 or A.shape[0] != A.shape[1] or B.shape[0] != A.shape[0] or B.shape[2] != A.shape[2]:
            raise ValueError("A must be (nx,nx,Bsz) and B (nx,nu,Bsz), instance-minor")
        return A, B, A.shape[0], B.shape[1], A.shape[2]

    def solve_batch(self, N, A, B, Q, R, P, lb, ub, x0, x_ref=None, u_ref=None):
        A, B, nx, nu, Bsz = self._dims(A, B)
        Q, R, P = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx))
        lb, ub, x0 = _f64(lb, (nu,)), _f64(ub, (nu,)), _f64(x0, (nx, Bsz))
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        u0 = np.empty((nu, Bsz)); VN = np.empty(Bsz)
        status = np.empty(Bsz, dtype=np.int32); iters = np.empty(Bsz, dtype=np.int32)
        _lib.check(self._L.lqmpc_solve_batch(self._h, nx, nu, N, Bsz, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(P),
                                             _ptr(lb), _ptr(ub), _ptr(x0), _ptr(x_ref), _ptr(u_ref),
                                             _ptr(u0), _ptr(VN), _ptr(status), _ptr(iters)))
        return {"u_0": u0, "V_N": VN, "status": status, "iters": iters}

    def rollout_batch(self, T, N, A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref=None, u_ref=None, want_traj=False):
        A, B, nx, nu, Bsz = self._dims(A, B)
        Q, R, P = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx))
        lb, ub, x0 = _f64(lb, (nu,)), _f64(ub, (nu,)), _f64(x0, (nx, Bsz))
        A_true, B_true = _f64(A_true), _f64(B_true)
        per_inst = 1 if A_true.ndim == 3 else 0
        if per_inst:
            A_true, B_true = _f64(A_true, (nx, nx, Bsz)), _f64(B_true, (nx, nu, Bsz))
        else:
            A_true, B_true = _f64(A_true, (nx, nx)), _f64(B_true, (nx, nu))
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        JT = np.empty(Bsz)
        X = np.empty((nx, T + 1, Bsz)) if want_traj else None
        U = np.empty((nu, T, Bsz)) if want_traj else None
        status = np.empty(Bsz, dtype=np.int32); iters = np.empty(Bsz, dtype=np.int32)
        _lib.check(self._L.lqmpc_rollout_batch(self._h, nx, nu, N, Bsz, T, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(P),
                                               _ptr(lb), _ptr(ub), _ptr(x0), _ptr(A_true), _ptr(B_true), per_inst,
                                               _ptr(x_ref), _ptr(u_ref), _ptr(JT), _ptr(X), _ptr(U),
                                               _ptr(status), _ptr(iters)))
        return {"J_T": JT, "X": X, "U": U, "status": status, "iters": iters}

    def max_vn_batch(self, N, A, B, Q, R, P, lb, ub, x0s, x_ref=None, u_ref=None):
        A, B, nx, nu, Bsz = self._dims(A, B)
        Q, R, P = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx))
        lb, ub = _f64(lb, (nu,)), _f64(ub, (nu,))
        x0s = _f64(x0s)
        if x0s.ndim != 2 or x0s.shape[0] != nx:
            raise ValueError("x0s must be (nx, K)")
        K = x0s.shape[1]
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        MV = np.empty(Bsz)
        status = np.empty(Bsz, dtype=np.int32); iters = np.empty(Bsz, dtype=np.int32)
        _lib.check(self._L.lqmpc_max_vn_batch(self._h, nx, nu, N, Bsz, K, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(P),
                                              _ptr(lb), _ptr(ub), _ptr(x0s), _ptr(x_ref), _ptr(u_ref),
                                              _ptr(MV), _ptr(status), _ptr(iters)))
        return {"M_V": MV, "status": status, "iters": iters}

    def sweep_batch(self, T, N, A, B, Q, R, P, lb, ub, x0, x0s, A_true, B_true, x_ref=None, u_ref=None):
        """One horizon of the reference's sweep (utils_class.py:813-833): M_V over the K columns of x0s and the closed-loop
        cost J_T from x0, for the same models, in one call (one launch where the 16-lane-row layout serves the shape)."""
        A, B, nx, nu, Bsz = self._dims(A, B)
        Q, R, P = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx))
        lb, ub, x0 = _f64(lb, (nu,)), _f64(ub, (nu,)), _f64(x0, (nx, Bsz))
        x0s = _f64(x0s)
        if x0s.ndim != 2 or x0s.shape[0] != nx:
            raise ValueError("x0s must be (nx, K)")
        A_true, B_true = _f64(A_true), _f64(B_true)
        per_inst = 1 if A_true.ndim == 3 else 0
        if per_inst:
            A_true, B_true = _f64(A_true, (nx, nx, Bsz)), _f64(B_true, (nx, nu, Bsz))
        else:
            A_true, B_true = _f64(A_true, (nx, nx)), _f64(B_true, (nx, nu))
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        JT = np.empty(Bsz); MV = np.empty(Bsz)
        status = np.empty(Bsz, dtype=np.int32); iters = np.empty(Bsz, dtype=np.int32)
        _lib.check(self._L.lqmpc_sweep_batch(self._h, nx, nu, N, Bsz, T, x0s.shape[1], _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(P),
                                             _ptr(lb), _ptr(ub), _ptr(x0), _ptr(x0s), _ptr(A_true), _ptr(B_true), per_inst,
                                             _ptr(x_ref), _ptr(u_ref), _ptr(JT), _ptr(MV), _ptr(status), _ptr(iters)))
        return {"J_T": JT, "M_V": MV, "status": status, "iters": iters}

    def bounds_batch(self, N, A, B, Q, R, lb, ub, e_A, e_B, M_V, x, p, V_expert, want_aux=False):
        """Per model of the batch: dlqr gain K and the coefficients alpha, beta, xi, eta, bound of the reference's performance
        bound (utils_class.py:837-859: control.dlqr + energy_decreasing + energy_bound), on the GPU (lqmpc_bounds_batch).
        e_A, e_B, M_V: (Bsz,) arrays (scalars are broadcast; M_V may be None = 0)."""
        A, B, nx, nu, Bsz = self._dims(A, B)
        Q, R, lb, ub = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(lb, (nu,)), _f64(ub, (nu,))
        e_A = np.ascontiguousarray(np.broadcast_to(np.asarray(e_A, dtype=np.float64), (Bsz,)))
        e_B = np.ascontiguousarray(np.broadcast_to(np.asarray(e_B, dtype=np.float64), (Bsz,)))
        M_V = None if M_V is None else np.ascontiguousarray(np.broadcast_to(np.asarray(M_V, dtype=np.float64), (Bsz,)))
        x, p = _f64(x, (nx,)), _f64(p, (3,))
        out = {k: np.empty(Bsz) for k in ("alpha", "beta", "xi", "eta", "bound", "eps")}
        out["K"] = np.empty((nu, nx, Bsz))
        out["status"] = np.empty(Bsz, dtype=np.int32)
        aux = np.empty((8, Bsz)) if want_aux else None
        _lib.check(self._L.lqmpc_bounds_batch(self._h, nx, nu, N, Bsz, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(lb), _ptr(ub),
                                              _ptr(e_A), _ptr(e_B), _ptr(M_V), _ptr(x), _ptr(p), float(V_expert),
                                              _ptr(out["K"]), _ptr(out["alpha"]), _ptr(out["beta"]), _ptr(out["xi"]),
                                              _ptr(out["eta"]), _ptr(out["bound"]), _ptr(out["eps"]), _ptr(aux), _ptr(out["status"])))
        if want_aux:
            out.update(dict(zip(("gamma", "rho_cl", "norm_A", "norm_B", "norm_Gamma", "norm_Phi", "min_eig_H", "norm_K"), aux)))
        return out

    # ---- device-pointer entry points (torch tensors / raw addresses already in HBM; asynchronous) ----
    def solve_batch_dev(self, nx, nu, N, Bsz, dA, dB, Q, R, P, lb, ub, dx0, du0, dVN, dstatus=None, diters=None,
                        x_ref=None, u_ref=None):
        Q, R, P, lb, ub = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx)), _f64(lb, (nu,)), _f64(ub, (nu,))
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        _lib.check(self._L.lqmpc_solve_batch_dev(self._h, nx, nu, N, Bsz, _ptr(dA), _ptr(dB), _ptr(Q), _ptr(R), _ptr(P),
                                                 _ptr(lb), _ptr(ub), _ptr(dx0), _ptr(x_ref), _ptr(u_ref),
                                                 _ptr(du0), _ptr(dVN), _ptr(dstatus), _ptr(diters)))

    def rollout_batch_dev(self, nx, nu, N, Bsz, T, dA, dB, Q, R, P, lb, ub, dx0, A_true, B_true, dJT,
                          dX=None, dU=None, dstatus=None, diters=None, true_per_instance=False, x_ref=None, u_ref=None):
        Q, R, P, lb, ub = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx)), _f64(lb, (nu,)), _f64(ub, (nu,))
        if not true_per_instance:
            A_true, B_true = _f64(A_true, (nx, nx)), _f64(B_true, (nx, nu))
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        _lib.check(self._L.lqmpc_rollout_batch_dev(self._h, nx, nu, N, Bsz, T, _ptr(dA), _ptr(dB), _ptr(Q), _ptr(R), _ptr(P),
                                                   _ptr(lb), _ptr(ub), _ptr(dx0), _ptr(A_true), _ptr(B_true),
                                                   1 if true_per_instance else 0, _ptr(x_ref), _ptr(u_ref),
                                                   _ptr(dJT), _ptr(dX), _ptr(dU), _ptr(dstatus), _ptr(diters)))

    def max_vn_batch_dev(self, nx, nu, N, Bsz, dA, dB, Q, R, P, lb, ub, x0s, dMV, dstatus=None, diters=None,
                         x_ref=None, u_ref=None):
        Q, R, P, lb, ub = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx)), _f64(lb, (nu,)), _f64(ub, (nu,))
        x0s = _f64(x0s)
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        _lib.check(self._L.lqmpc_max_vn_batch_dev(self._h, nx, nu, N, Bsz, x0s.shape[1], _ptr(dA), _ptr(dB), _ptr(Q), _ptr(R),
                                                  _ptr(P), _ptr(lb), _ptr(ub), _ptr(x0s), _ptr(x_ref), _ptr(u_ref),
                                                  _ptr(dMV), _ptr(dstatus), _ptr(diters)))

    def sweep_batch_dev(self, nx, nu, N, Bsz, T, dA, dB, Q, R, P, lb, ub, dx0, x0s, A_true, B_true, dJT, dMV,
                        dstatus=None, diters=None, true_per_instance=False, x_ref=None, u_ref=None):
        Q, R, P, lb, ub = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(P, (nx, nx)), _f64(lb, (nu,)), _f64(ub, (nu,))
        x0s = _f64(x0s)
        if x0s.ndim != 2 or x0s.shape[0] != nx:
            raise ValueError("x0s must be (nx, K)")
        if not true_per_instance:
            A_true, B_true = _f64(A_true, (nx, nx)), _f64(B_true, (nx, nu))
        x_ref, u_ref = _ref_or_none(x_ref, nx, N), _ref_or_none(u_ref, nu, N)
        _lib.check(self._L.lqmpc_sweep_batch_dev(self._h, nx, nu, N, Bsz, T, x0s.shape[1], _ptr(dA), _ptr(dB), _ptr(Q), _ptr(R),
                                                 _ptr(P), _ptr(lb), _ptr(ub), _ptr(dx0), _ptr(x0s), _ptr(A_true), _ptr(B_true),
                                                 1 if true_per_instance else 0, _ptr(x_ref), _ptr(u_ref),
                                                 _ptr(dJT), _ptr(dMV), _ptr(dstatus), _ptr(diters)))

    def bounds_batch_dev(self, nx, nu, N, Bsz, dA, dB, Q, R, lb, ub, de_A, de_B, dMV, x, p, V_expert, dK=None, dalpha=None,
                         dbeta=None, dxi=None, deta=None, dbound=None, deps=None, daux=None, dstatus=None):
        Q, R, lb, ub = _f64(Q, (nx, nx)), _f64(R, (nu, nu)), _f64(lb, (nu,)), _f64(ub, (nu,))
        x, p = _f64(x, (nx,)), _f64(p, (3,))
        _lib.check(self._L.lqmpc_bounds_batch_dev(self._h, nx, nu, N, Bsz, _ptr(dA), _ptr(dB), _ptr(Q), _ptr(R), _ptr(lb), _ptr(ub),
                                                  _ptr(de_A), _ptr(de_B), _ptr(dMV), _ptr(x), _ptr(p), float(V_expert),
                                                  _ptr(dK), _ptr(dalpha), _ptr(dbeta), _ptr(dxi), _ptr(deta), _ptr(dbound),
                                                  _ptr(deps), _ptr(daux), _ptr(dstatus)))


_default_solver = None


def default_solver():
    """Process-wide solver on GPU 0 used by the single-instance classes below."""
    global _default_solver
    if _default_solver is None:
        _default_solver = BatchSolver(0)
    return _default_solver


class LQ_MPC_Controller:
    """Open-loop LQ MPC solve; drop-in for utils_class.py:18-91 (box-shaped F_u only)."""

    def __init__(self, N, A, B, Q, R, P, F_u, solver=None):
        self.N = int(N)
        self.A = np.asarray(A, dtype=np.float64)
        self.B = np.asarray(B, dtype=np.float64)
        self.Q = np.asarray(Q, dtype=np.float64)
        self.R = np.asarray(R, dtype=np.float64)
        self.P = np.asarray(P, dtype=np.float64)
        self.F_u = np.asarray(F_u, dtype=np.float64)
        self.lb, self.ub = box_from_Fu(self.F_u)
        self._solver = solver

    def _s(self):
        return self._solver if self._solver is not None else default_solver()

    def solve(self, x0, x_ref, u_ref):
        """Returns {'u_0': (nu,) ndarray, 'V_N': float} as utils_class.py:91."""
        nx = self.A.shape[0]
        x0 = np.asarray(x0, dtype=np.float64).reshape(nx, 1)
        out = self._s().solve_batch(self.N, self.A[:, :, None], self.B[:, :, None], self.Q, self.R, self.P,
                                    self.lb, self.ub, x0, x_ref, u_ref)
        self.last_status = int(out["status"][0])
        return {"u_0": out["u_0"][:, 0].copy(), "V_N": float(out["V_N"][0])}

    def solve_many(self, x0s, x_ref=None, u_ref=None):
        """Additive: K initial states of one system in one launch -> {'u_0': (nu,K), 'V_N': (K,)}."""
        x0s = np.asarray(x0s, dtype=np.float64)
        K = x0s.shape[1]
        A = np.repeat(self.A[:, :, None], K, 2)
        B = np.repeat(self.B[:, :, None], K, 2)
        out = self._s().solve_batch(self.N, A, B, self.Q, self.R, self.P, self.lb, self.ub, x0s, x_ref, u_ref)
        return {"u_0": out["u_0"], "V_N": out["V_N"], "status": out["status"]}


class LQ_MPC_Simulator:
    """Closed-loop MPC rollout; drop-in for utils_class.py:213-285."""

    def __init__(self, T, N, A, B, Q, R, P, F_u, solver=None):
        self.T = int(T)
        self.N = int(N)
        self.A = np.asarray(A, dtype=np.float64)
        self.B = np.asarray(B, dtype=np.float64)
        self.Q = np.asarray(Q, dtype=np.float64)
        self.R = np.asarray(R, dtype=np.float64)
        self.P = np.asarray(P, dtype=np.float64)
        self.F_u = np.asarray(F_u, dtype=np.float64)
        self.lb, self.ub = box_from_Fu(self.F_u)
        self.U = np.zeros((self.B.shape[1], self.T))
        self.X = np.zeros((self.A.shape[1], self.T + 1))
        self._solver = solver

    def simulate(self, x0, A_true, B_true, x_ref, u_ref):
        """Returns {'X': (nx,T+1), 'U': (nu,T), 'J_T': float} as utils_class.py:285 (fresh arrays)."""
        s = self._solver if self._solver is not None else default_solver()
        nx = self.A.shape[0]
        x0 = np.asarray(x0, dtype=np.float64).reshape(nx, 1)
        out = s.rollout_batch(self.T, self.N, self.A[:, :, None], self.B[:, :, None], self.Q, self.R, self.P,
                              self.lb, self.ub, x0, np.asarray(A_true, dtype=np.float64),
                              np.asarray(B_true, dtype=np.float64), x_ref, u_ref, want_traj=True)
        self.X = out["X"][:, :, 0].copy()
        self.U = out["U"][:, :, 0].copy()
        self.last_status = int(out["status"][0])
        return {"X": self.X, "U": self.U, "J_T": float(out["J_T"][0])}
