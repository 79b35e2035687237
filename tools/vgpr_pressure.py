"""Dev tool: an estimate of the live vector registers at every instruction of a kernel in a gfx950 assembly listing (hipcc -S), by a
backward data-flow over its basic blocks.  Prints the peak and the pressure at every label, so the region that sets the register
count of a kernel shows.  usage: vgpr_pressure.py file.s kernel_symbol [min_live_to_print]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
sym = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 0
start = next(i for i, l in enumerate(txt) if l.startswith(sym + ':'))
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
ins = []          # (line, label or None, opcode, operands text)
labels = {}
for i in range(start + 1, end):
    l = txt[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    if not l.startswith('\t') or l.startswith('\t.') or l.startswith('\t;'): continue
    body = l.split(';')[0].strip()
    if not body: continue
    parts = body.split(None, 1)
    ins.append((i + 1, parts[0], parts[1] if len(parts) > 1 else ''))
def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok): out.update(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok): out.add(int(a))
    return out
NODEF = ('global_store', 'flat_store', 'ds_write', 'buffer_store', 'scratch_store', 'v_cmp', 'v_readlane', 'v_readfirstlane', 's_', 'ds_bpermute_none', 'global_atomic')
def defs_uses(op, args):
    toks = [t.strip() for t in args.split(',')] if args else []
    if not toks: return set(), set()
    nodef = op.startswith(NODEF) and not (op.startswith('global_atomic') and 'sc0' in args)
    d = set() if nodef else regs(toks[0])
    u = set()
    for t in (toks if nodef else toks[1:]): u |= regs(t)
    if op.startswith(('v_fmac', 'v_mac', 'v_dot')) or '_dpp' in op or 'sdwa' in op or op.startswith('v_mov_b32_dpp') or op.startswith('v_cndmask') is False and False:
        u |= d            # (read-modify-write: the DPP forms keep the old value where no lane supplies one)
    if op.startswith('v_writelane'): u |= d
    return d, u
n = len(ins)
succ = [[] for _ in range(n)]
for k, (ln, op, args) in enumerate(ins):
    if op.startswith('s_branch'):
        t = args.strip()
        if t in labels: succ[k].append(labels[t])
    elif op.startswith('s_cbranch'):
        t = args.strip()
        if t in labels: succ[k].append(labels[t])
        if k + 1 < n: succ[k].append(k + 1)
    elif op.startswith('s_endpgm'): pass
    elif op.startswith('s_setpc') or op.startswith('s_swappc'):
        if k + 1 < n: succ[k].append(k + 1)
    elif k + 1 < n: succ[k].append(k + 1)
DU = [defs_uses(op, args) for (_, op, args) in ins]
live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for k in range(n - 1, -1, -1):
        out = set()
        for s in succ[k]: out |= live_in[s]
        d, u = DU[k]
        new = (out - d) | u
        if new != live_in[k]: live_in[k] = new; changed = True
peak = max(range(n), key=lambda k: len(live_in[k]))
print('instructions', n, ' peak live vector registers', len(live_in[peak]), 'at line', ins[peak][0], ins[peak][1])
inv = {v: k for k, v in labels.items()}
step = max(1, n // 120)
for k in range(n):
    if (k in inv or k % step == 0) and len(live_in[k]) >= thr:
        print('%6d %-12s live %3d  %s %s' % (ins[k][0], inv.get(k, ''), len(live_in[k]), ins[k][1], ins[k][2][:50]))
