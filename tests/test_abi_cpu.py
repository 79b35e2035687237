"""CPU: the C-ABI library loads and exports every symbol include/lqmpc.h declares; host logic."""
import ctypes
import os
import re

import numpy as np
import pytest

import lq_mpc_amd
from lq_mpc_amd import _lib, box_from_Fu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "lqmpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lqmpc_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    assert sorted(_lib.EXPORTS) == declared
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_version_and_defaults():
    L = _lib.lib()
    assert b"gfx950" in L.lqmpc_version()
    o = _lib.Options()
    L.lqmpc_default_options(ctypes.byref(o))
    assert o.eps == 1e-12 and o.max_iter == 50 and o.polish == 1 and o.kernel == _lib.KERNEL_AUTO and o.presolve == -1 and o.order == -1 and o.warm_start == -1
    assert o.layout == -1 and o.r16_maxit == 12 and o.r16_build == -1 and o.nwide == -1
    assert ctypes.sizeof(_lib.Options) == 80 and o.struct_size == 80 and o.reserved == 0 and o.jit == -1     # the struct carries its own size (ABI 0.2.0)


def test_no_device_is_an_error_not_a_fallback():
    L = _lib.lib()
    if L.lqmpc_device_count() > 0:
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = L.lqmpc_create(0, ctypes.byref(h))
    assert rc == -3 and not h.value and b"no HIP device" in L.lqmpc_last_error()
    with pytest.raises(_lib.LqmpcError):
        lq_mpc_amd.BatchSolver(0)


def test_bad_handle_arguments():
    L = _lib.lib()
    assert L.lqmpc_sync(None) == -1
    assert L.lqmpc_destroy(None) == 0
    assert L.lqmpc_timer_begin(None) == -1


def test_box_from_Fu():
    lb, ub = box_from_Fu(np.vstack((10 * np.eye(2), -10 * np.eye(2))))      # working_example_multiple.py:25
    np.testing.assert_allclose(lb, [-0.1, -0.1]); np.testing.assert_allclose(ub, [0.1, 0.1])
    lb, ub = box_from_Fu(np.array([[4.0, 0], [0, 2.0], [-5.0, 0], [0, -1.0], [8.0, 0]]))
    np.testing.assert_allclose(lb, [-0.2, -1.0]); np.testing.assert_allclose(ub, [0.125, 0.5])
    with pytest.raises(ValueError):
        box_from_Fu(np.array([[1.0, 1.0], [-1.0, 0.0]]))         # not a box row
    with pytest.raises(ValueError):
        box_from_Fu(np.array([[1.0, 0.0], [0.0, 1.0]]))          # unbounded below


def test_product_path_does_not_import_oracle():
    src_dir = os.path.join(ROOT, "lq_mpc_amd")
    for dirpath, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} mentions the oracle"


def test_sweep_host_helpers_match_reference_constants():
    """circle_generator restated for the product's sweep driver (SURVEY 8(c) constants; K and eps come from the GPU there,
    tests/test_gpu_bounds.py)."""
    from lq_mpc_amd import sweep
    Q = 2.0 * np.eye(2)
    eps = 0.04503580745099056
    x0 = sweep.circle_generator(8, 1.5, eps, Q)
    np.testing.assert_allclose(x0[:, 1], [0.15916231240837822, 0.15916231240837819], rtol=1e-14)
    # n_x > 2: same level set, seeded planes
    Q4 = np.diag([1.0, 2.0, 3.0, 4.0]) + 0.1
    x4 = sweep.circle_generator(8, 1.3, 0.2, Q4)
    np.testing.assert_allclose(np.einsum("ik,ij,jk->k", x4, Q4, x4), 1.3 ** 2 * 0.2, rtol=1e-13)
    assert np.linalg.matrix_rank(x4) == 4 and np.array_equal(x4, sweep.circle_generator(8, 1.3, 0.2, Q4))
    assert sweep.circle_generator(5, 1.0, 1.0, np.eye(3)).shape == (3, 5)


def test_references_wider_than_the_horizon_are_accepted():
    """utils_class.py:69, 75 read x_ref[:, i], u_ref[:, i] for i < N only."""
    from lq_mpc_amd.mpc import _ref_or_none
    r = np.arange(12.0).reshape(2, 6)
    np.testing.assert_array_equal(_ref_or_none(r, 2, 4), r[:, :4])
    assert _ref_or_none(np.zeros((2, 9)), 2, 4) is None and _ref_or_none(None, 2, 4) is None
    with pytest.raises(ValueError):
        _ref_or_none(r, 2, 7)
    with pytest.raises(ValueError):
        _ref_or_none(r, 3, 4)
