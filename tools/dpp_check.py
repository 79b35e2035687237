"""Dev / test tool: the gfx950 DPP read-after-write rule, checked on the disassembly of the shipped code objects.

A DPP instruction reads its src0 through the cross-lane network; a VGPR written by a VALU instruction needs two wait states before
a DPP read of it.  The compiler inserts them for its own instructions, but it does not look into inline asm: the hand-written
`v_fmac_f64_dpp` statements of lqmpc_wg_linalg.h carry their own `s_nop 1`, and nothing but this check would notice if a compiler
update moved a copy between that s_nop and the read.  For every *_dpp instruction: walk back over the preceding instructions of the
same function, counting wait states (s_nop N = N + 1, any other instruction = 1); a VALU write of a src0 register met before two
wait states have passed is a violation.

usage: python tools/dpp_check.py [liblqmpc_hip.so | code objects ...]   (exit code 1 on a violation)"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(op):
    out = set()
    for m in _REG.finditer(op):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def code_objects(path):
    """gfx950 code objects inside a host shared library (offload bundles), or the file itself if it already is one."""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[18:20] == b"\xe0\x00":                       # e_machine = EM_AMDGPU (224)
        return [path], None
    tmp = tempfile.mkdtemp(prefix="lqmpc_dpp_")
    local = os.path.join(tmp, os.path.basename(path))
    os.symlink(os.path.abspath(path), local)
    subprocess.run([OBJDUMP, "--offloading", os.path.basename(path)], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f and os.path.getsize(os.path.join(tmp, f)) > 0), tmp


def check_listing(lines, where):
    """lines: llvm-objdump -d output.  Returns (dpp instructions checked, violations)."""
    checked, bad = 0, []
    hist = []                                            # (mnemonic, operand string) of the current function
    func = "?"
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            func, hist = m.group(1), []
            continue
        if not ln.startswith("\t"):
            continue
        txt = ln.split("//")[0].strip()
        if not txt:
            continue
        parts = txt.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        if mn.endswith("_dpp"):
            checked += 1
            oplist = [o.strip() for o in ops.split(",")]
            src0 = _regs(oplist[1].split()[0]) if len(oplist) > 1 else set()
            wait = 0
            for pmn, pops in reversed(hist):
                if wait >= 2:
                    break
                if pmn == "s_nop":
                    wait += int(pops.split()[0], 0) + 1
                    continue
                if pmn.startswith("v_") and not pmn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_accvgpr_write")):
                    dst = _regs(pops.split(",")[0])
                    if pmn.startswith("v_swap"):
                        dst |= _regs(pops.split(",")[1])
                    if dst & src0:
                        bad.append(f"{where}: {func}: `{txt}` reads v{sorted(dst & src0)} through DPP {wait} wait state(s) after `{pmn} {pops}`")
                        break
                wait += 1
        hist.append((mn, ops))
        if len(hist) > 8:
            hist.pop(0)
    return checked, bad


def check_file(path):
    objs, tmp = code_objects(path)
    total, bad = 0, []
    for o in objs:
        out = subprocess.run([OBJDUMP, "-d", o], capture_output=True, text=True).stdout.splitlines()
        c, b = check_listing(out, os.path.basename(o))
        total += c
        bad += b
    return total, bad


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sys.argv[1:] or [os.path.join(root, "lq_mpc_amd", "liblqmpc_hip.so")]
    n, bad = 0, []
    for f in files:
        c, b = check_file(f)
        n += c
        bad += b
    print(f"{n} DPP instructions checked, {len(bad)} violation(s)")
    for b in bad[:50]:
        print(b)
    sys.exit(1 if bad else 0)
