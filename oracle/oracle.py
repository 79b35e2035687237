"""ctypes front-end of the CPU oracle (oracle/lqmpc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under lq_mpc_amd/ imports this module.

Besides the C entry points this file restates, in numpy, the tiny host-side input
preparation the reference performs before it reaches the hot path, so the golden test can
regenerate the reference's own inputs:
  * circle_generator        /root/reference/utils.py:683-704 (+ rot_2D 658-680)
  * local_radius            /root/reference/utils.py:548-564
  * dlqr gain (ct.dlqr)     /root/reference/utils_class.py:761  (scipy DARE instead of `control`)
  * F_u -> (lb, ub)         /root/reference/utils_class.py:81   (box rows only)
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblqmpc_oracle.so")
_lib = None

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the oracle with gcc (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = ctypes.CDLL(_LIB_PATH)
        except OSError:
            build(force=True)
            _lib = ctypes.CDLL(_LIB_PATH)
        _lib.lqo_num_threads.restype = ctypes.c_int
    return _lib


def use_native_build():
    """bench.py's cpu_baseline leg: rebuild the oracle with -march=native for the host it is timed on (into a temp directory;
    the shipped library stays as it is) and switch this module to it.  Returns False (and keeps the shipped build) when no
    compiler is available."""
    global _lib
    import tempfile
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="lqmpc_oracle_native_"), "liblqmpc_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", out,
                               os.path.join(_HERE, "lqmpc_oracle.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nat = ctypes.CDLL(out)
        nat.lqo_num_threads.restype = ctypes.c_int
        _lib = nat
        return True
    except Exception:
        return False


def _p(a):
    return None if a is None else a.ctypes.data_as(_D)


def _c(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def num_threads():
    return int(lib().lqo_num_threads())


def box_from_Fu(F_u):
    """Rows of F_u with exactly one non-zero -> per-input bounds (utils_class.py:81)."""
    F_u = np.atleast_2d(np.asarray(F_u, dtype=np.float64))
    nu = F_u.shape[1]
    lb = np.full(nu, -np.inf)
    ub = np.full(nu, np.inf)
    for row in F_u:
        nz = np.flatnonzero(row)
        if nz.size != 1:
            raise ValueError("only box constraints (one non-zero per row of F_u) are supported")
        k = nz[0]
        if row[k] > 0:
            ub[k] = min(ub[k], 1.0 / row[k])
        else:
            lb[k] = max(lb[k], 1.0 / row[k])
    return lb, ub


def condense(A, B, Q, R, P, N):
    A, B, Q, R, P = map(_c, (A, B, Q, R, P))
    nx, nu = B.shape
    n = N * nu
    H = np.zeros((n, n))
    F = np.zeros((n, nx))
    rc = lib().lqo_condense(nx, nu, N, _p(A), _p(B), _p(Q), _p(R), _p(P), _p(H), _p(F))
    if rc:
        raise RuntimeError(f"lqo_condense rc={rc}")
    return H, F


def boxqp(H, g, lb, ub):
    """Exact argmin u'Hu + 2g'u on the box; returns (u, active-set iterations)."""
    H, g, lb, ub = map(_c, (H, g, lb, ub))
    n = g.size
    u = np.zeros(n)
    rc = lib().lqo_boxqp(n, _p(H), _p(g), _p(lb), _p(ub), _p(u))
    if rc < 0:
        raise RuntimeError(f"lqo_boxqp rc={rc}")
    return u, rc


def solve(N, A, B, Q, R, P, lb, ub, x0, x_ref=None, u_ref=None):
    """Mirror of LQ_MPC_Controller(N,A,B,Q,R,P,F_u).solve(x0,x_ref,u_ref) -> dict with U too."""
    A, B, Q, R, P, lb, ub, x0, x_ref, u_ref = map(_c, (A, B, Q, R, P, lb, ub, x0, x_ref, u_ref))
    nx, nu = B.shape
    u0 = np.zeros(nu)
    U = np.zeros(N * nu)
    VN = ctypes.c_double()
    rc = lib().lqo_solve(nx, nu, N, _p(A), _p(B), _p(Q), _p(R), _p(P), _p(lb), _p(ub), _p(x0),
                         _p(x_ref), _p(u_ref), _p(u0), ctypes.byref(VN), _p(U))
    if rc < 0:
        raise RuntimeError(f"lqo_solve rc={rc}")
    return {"u_0": u0, "V_N": VN.value, "U": U.reshape(N, nu).T.copy(), "iters": rc}


def simulate(T, N, A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref=None, u_ref=None):
    """Mirror of LQ_MPC_Simulator(T,N,A,B,Q,R,P,F_u).simulate(x0,A_true,B_true,x_ref,u_ref)."""
    A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref, u_ref = map(
        _c, (A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref, u_ref))
    nx, nu = B.shape
    X = np.zeros((nx, T + 1))
    U = np.zeros((nu, T))
    JT = ctypes.c_double()
    rc = lib().lqo_simulate(T, nx, nu, N, _p(A), _p(B), _p(Q), _p(R), _p(P), _p(lb), _p(ub), _p(x0),
                            _p(A_true), _p(B_true), _p(x_ref), _p(u_ref), ctypes.byref(JT), _p(X), _p(U))
    if rc < 0:
        raise RuntimeError(f"lqo_simulate rc={rc}")
    return {"X": X, "U": U, "J_T": JT.value}


# ---- batched (SoA, instance-minor: A (nx,nx,Bsz), B (nx,nu,Bsz), x0 (nx,Bsz)) ----
def solve_batch(N, A, B, Q, R, P, lb, ub, x0, x_ref=None, u_ref=None, threads=0):
    A, B, Q, R, P, lb, ub, x0, x_ref, u_ref = map(_c, (A, B, Q, R, P, lb, ub, x0, x_ref, u_ref))
    nx, nu, Bsz = B.shape
    u0 = np.zeros((nu, Bsz))
    VN = np.zeros(Bsz)
    iters = np.zeros(Bsz, dtype=np.int32)
    rc = lib().lqo_solve_batch(ctypes.c_long(Bsz), nx, nu, N, _p(A), _p(B), _p(Q), _p(R), _p(P), _p(lb), _p(ub),
                               _p(x0), _p(x_ref), _p(u_ref), _p(u0), _p(VN),
                               iters.ctypes.data_as(_I), int(threads))
    if rc:
        raise RuntimeError(f"lqo_solve_batch rc={rc}")
    return {"u_0": u0, "V_N": VN, "iters": iters}


def rollout_batch(T, N, A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref=None, u_ref=None,
                  want_traj=False, threads=0):
    A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref, u_ref = map(
        _c, (A, B, Q, R, P, lb, ub, x0, A_true, B_true, x_ref, u_ref))
    nx, nu, Bsz = B.shape
    shared = 1 if A_true.ndim == 2 else 0
    JT = np.zeros(Bsz)
    X = np.zeros((nx, T + 1, Bsz)) if want_traj else None
    U = np.zeros((nu, T, Bsz)) if want_traj else None
    rc = lib().lqo_rollout_batch(ctypes.c_long(Bsz), T, nx, nu, N, _p(A), _p(B), _p(Q), _p(R), _p(P),
                                 _p(lb), _p(ub), _p(x0), _p(A_true), _p(B_true), shared,
                                 _p(x_ref), _p(u_ref), _p(JT), _p(X), _p(U), int(threads))
    if rc:
        raise RuntimeError(f"lqo_rollout_batch rc={rc}")
    return {"J_T": JT, "X": X, "U": U}


def max_vn_batch(N, A, B, Q, R, P, lb, ub, x0s, x_ref=None, u_ref=None, threads=0):
    A, B, Q, R, P, lb, ub, x0s, x_ref, u_ref = map(_c, (A, B, Q, R, P, lb, ub, x0s, x_ref, u_ref))
    nx, nu, Bsz = B.shape
    K = x0s.shape[1]
    MV = np.zeros(Bsz)
    rc = lib().lqo_max_vn_batch(ctypes.c_long(Bsz), K, nx, nu, N, _p(A), _p(B), _p(Q), _p(R), _p(P),
                                _p(lb), _p(ub), _p(x0s), _p(x_ref), _p(u_ref), _p(MV), int(threads))
    if rc:
        raise RuntimeError(f"lqo_max_vn_batch rc={rc}")
    return MV


# ---- numpy restatements of the reference's input preparation (see module docstring) ----
def dlqr_gain(A, B, Q, R):
    """K of ct.dlqr (convention u = -K x), via scipy's DARE."""
    from scipy.linalg import solve_discrete_are
    Pinf = solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(R + B.T @ Pinf @ B, B.T @ Pinf @ A)
    return K, Pinf


def local_radius(F_u, K, Q):
    M = F_u @ K
    invQ = np.linalg.inv(Q)
    a = np.array([M[i] @ invQ @ M[i] for i in range(M.shape[0])])
    return 1.0 / np.max(a)


def circle_generator(N_points, ratio_ext_radius, my_base, Q):
    """n_x = 2 only, as the reference; cho_factor's default factor is UPPER."""
    root_Q = np.linalg.cholesky(Q).T
    x0_base = np.array([ratio_ext_radius * math.sqrt(my_base), 0.0])
    theta = np.linspace(0, 2 * (1 - 1 / N_points) * math.pi, N_points)
    pts = np.zeros((2, N_points))
    for i, th in enumerate(theta):
        Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        pts[:, i] = Rm @ x0_base
    return np.linalg.inv(root_Q) @ pts
