"""ctypes binding of liblqmpc_hip.so (the C ABI in include/lqmpc.h).

The product path has no CPU fallback: if the HIP library cannot be loaded, or no GPU is
visible when a solver handle is requested, this module raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblqmpc_hip.so")
JIT_CACHE = os.path.join(_HERE, "_jit_cache")
CSRC = os.path.join(_HERE, "csrc")

_D = ctypes.POINTER(ctypes.c_double)
_I32 = ctypes.POINTER(ctypes.c_int32)
_H = ctypes.c_void_p

# every symbol include/lqmpc.h declares
EXPORTS = [
    "lqmpc_version", "lqmpc_last_error", "lqmpc_device_count",
    "lqmpc_create", "lqmpc_create_on_stream", "lqmpc_destroy", "lqmpc_sync",
    "lqmpc_default_options", "lqmpc_set_options", "lqmpc_get_options",
    "lqmpc_has_specialization", "lqmpc_last_kernel", "lqmpc_reserve",
    "lqmpc_solve_batch", "lqmpc_solve_batch_dev",
    "lqmpc_rollout_batch", "lqmpc_rollout_batch_dev",
    "lqmpc_max_vn_batch", "lqmpc_max_vn_batch_dev",
    "lqmpc_sweep_batch", "lqmpc_sweep_batch_dev",
    "lqmpc_bounds_batch", "lqmpc_bounds_batch_dev",
    "lqmpc_timer_begin", "lqmpc_timer_end",
    "lqmpc_jit_cache_dir", "lqmpc_jit_compile", "lqmpc_jit_compile_bounds",
]


class Options(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("eps", ctypes.c_double), ("tau", ctypes.c_double), ("z0_scale", ctypes.c_double),
                ("max_iter", ctypes.c_int32), ("polish", ctypes.c_int32), ("kernel", ctypes.c_int32),
                ("presolve", ctypes.c_int32), ("order", ctypes.c_int32), ("warm_start", ctypes.c_int32),
                ("layout", ctypes.c_int32), ("r16_maxit", ctypes.c_int32), ("r16_build", ctypes.c_int32),
                ("nwide", ctypes.c_int32), ("jit", ctypes.c_int32), ("reserved2", ctypes.c_int32)]


KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALIZED, KERNEL_WORKGROUP = 0, 1, 2, 3


class LqmpcError(RuntimeError):
    pass


def build(force=False):
    """hipcc build of the shared library for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = ctypes.CDLL(LIB_PATH)
    L.lqmpc_version.restype = ctypes.c_char_p
    L.lqmpc_last_error.restype = ctypes.c_char_p
    L.lqmpc_last_kernel.restype = ctypes.c_char_p
    L.lqmpc_last_kernel.argtypes = [_H]
    L.lqmpc_device_count.restype = ctypes.c_int
    L.lqmpc_create.argtypes = [ctypes.c_int, ctypes.POINTER(_H)]
    L.lqmpc_create_on_stream.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(_H)]
    L.lqmpc_destroy.argtypes = [_H]
    L.lqmpc_sync.argtypes = [_H]
    L.lqmpc_default_options.argtypes = [ctypes.POINTER(Options)]
    L.lqmpc_default_options.restype = None
    L.lqmpc_set_options.argtypes = [_H, ctypes.POINTER(Options)]
    L.lqmpc_get_options.argtypes = [_H, ctypes.POINTER(Options)]
    L.lqmpc_has_specialization.argtypes = [ctypes.c_int] * 3
    L.lqmpc_reserve.argtypes = [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    P = ctypes.c_void_p   # data pointers: host (numpy) or device (raw address), as void*
    dims = [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
    solve_args = dims + [P] * 14
    L.lqmpc_solve_batch.argtypes = solve_args
    L.lqmpc_solve_batch_dev.argtypes = solve_args
    roll_args = dims + [ctypes.c_int] + [P] * 10 + [ctypes.c_int] + [P] * 7
    L.lqmpc_rollout_batch.argtypes = roll_args
    L.lqmpc_rollout_batch_dev.argtypes = roll_args
    mv_args = dims + [ctypes.c_int] + [P] * 13
    L.lqmpc_max_vn_batch.argtypes = mv_args
    L.lqmpc_max_vn_batch_dev.argtypes = mv_args
    sweep_args = dims + [ctypes.c_int, ctypes.c_int] + [P] * 11 + [ctypes.c_int] + [P] * 6
    L.lqmpc_sweep_batch.argtypes = sweep_args
    L.lqmpc_sweep_batch_dev.argtypes = sweep_args
    bounds_args = dims + [P] * 11 + [ctypes.c_double] + [P] * 9
    L.lqmpc_bounds_batch.argtypes = bounds_args
    L.lqmpc_bounds_batch_dev.argtypes = bounds_args
    L.lqmpc_timer_begin.argtypes = [_H]
    L.lqmpc_timer_end.argtypes = [_H, ctypes.POINTER(ctypes.c_float)]
    L.lqmpc_jit_cache_dir.argtypes = [ctypes.c_char_p]
    L.lqmpc_jit_compile.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    L.lqmpc_jit_compile_bounds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    # code objects of run-time compiled shapes are kept next to the library (falls back to memory only if not writable)
    L.lqmpc_jit_cache_dir(JIT_CACHE.encode())
    _lib = L
    return L


def jit_compile(nx, nu, N):
    """Compile (or find in the cache) every kernel of one shape now; needs no GPU.  Returns the number of code objects."""
    log = ctypes.create_string_buffer(4096)
    rc = lib().lqmpc_jit_compile(int(nx), int(nu), int(N), log, len(log))
    if rc < 0:
        raise LqmpcError(f"lqmpc_jit_compile({nx},{nu},{N}) -> {rc}: {log.value.decode(errors='replace')}")
    return rc


def jit_compile_bounds(nx, nu, N):
    """The two on-chip kernels of bounds_batch for one shape, compiled (or found in the cache) now; needs no GPU."""
    log = ctypes.create_string_buffer(4096)
    rc = lib().lqmpc_jit_compile_bounds(int(nx), int(nu), int(N), log, len(log))
    if rc < 0:
        raise LqmpcError(f"lqmpc_jit_compile_bounds({nx},{nu},{N}) -> {rc}: {log.value.decode(errors='replace')}")
    return rc


def check(rc):
    if rc != 0:
        raise LqmpcError(f"lqmpc error {rc}: {lib().lqmpc_last_error().decode()}")
