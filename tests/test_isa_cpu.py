"""CPU: properties of the shipped gfx950 machine code that only show as wrong numbers on the GPU otherwise.

The hand-written `v_fmac_f64_dpp` statements (lqmpc_wg_linalg.h) rely on two wait states between a VALU write of a register and a
DPP read of it; the compiler does not look into inline asm, so the rule is held by an `s_nop 1` inside every statement and by
`dpp_settle()` at the call sites of the builtin broadcasts.  tools/dpp_check.py walks the disassembly of every code object of
liblqmpc_hip.so (and of the run-time compiled kernels in the cache) and fails on any DPP read that comes too early."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dpp_check  # noqa: E402

from lq_mpc_amd import _lib  # noqa: E402


def test_checker_sees_a_planted_hazard():
    ok = ["0000000000001000 <k>:", "\tv_mul_f64 v[4:5], v[0:1], v[2:3]  // 0: 0", "\ts_nop 1  // 0: 0",
          "\tv_fmac_f64_dpp v[6:7], v[4:5], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf  // 0: 0"]
    assert dpp_check.check_listing(ok, "t") == (1, [])
    one_slot = [ok[0], ok[1], "\tv_add_u32_e32 v20, v21, v22  // 0: 0", ok[3]]
    n, bad = dpp_check.check_listing(one_slot, "t")
    assert n == 1 and len(bad) == 1 and "1 wait state" in bad[0]
    two_slots = [ok[0], ok[1], one_slot[2], one_slot[2], ok[3]]
    assert dpp_check.check_listing(two_slots, "t") == (1, [])
    mov = [ok[0], "\tv_mov_b32_e32 v3, v9  // 0: 0", "\tv_mov_b32_dpp v1, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf  // 0: 0"]
    assert len(dpp_check.check_listing(mov, "t")[1]) == 1
    other_reg = [ok[0], "\tv_mov_b32_e32 v30, v9  // 0: 0", mov[2]]
    assert dpp_check.check_listing(other_reg, "t") == (1, [])


def test_no_dpp_read_after_write_hazard_in_the_shipped_library():
    _lib.lib()                                            # (builds the library if it is not there)
    n, bad = dpp_check.check_file(_lib.LIB_PATH)
    assert n > 10000, "the library's code objects were not found or hold no DPP instructions"
    assert not bad, "\n".join(bad[:20])


def test_no_dpp_hazard_in_the_run_time_compiled_kernels():
    objs = sorted(glob.glob(os.path.join(_lib.JIT_CACHE, "*.hsaco")))[:12]      # (a sample: each takes a fraction of a second)
    if not objs:
        _lib.jit_compile(3, 2, 6)
        objs = sorted(glob.glob(os.path.join(_lib.JIT_CACHE, "*.hsaco")))[:12]
    n, bad = 0, []
    for o in objs:
        c, b = dpp_check.check_file(o)
        n += c
        bad += b
    assert n > 100 and not bad, "\n".join(bad[:20])
