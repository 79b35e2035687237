"""Dev tool: per-basic-block instruction mix of the gfx950 assembly of lqmpc_spec kernels."""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
minb = int(sys.argv[2]) if len(sys.argv) > 2 else 300
name = None; blocks = []; cur = None
def flush():
    global cur
    if cur: blocks.append(cur)
for l in txt:
    m = re.match(r'^(_ZN5lqmpc\S+):', l)
    if m:
        flush(); name = m.group(1); cur = [name + ' entry', 0, 0, 0, 0, 0, 0, 0]; continue
    if l.startswith('.Lfunc_end'):
        flush(); cur = None; name = None; continue
    if cur is None: continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        flush(); cur = [m.group(1), 0, 0, 0, 0, 0, 0, 0]; continue
    if not l.startswith('\t') or l.startswith('\t.') or l.startswith('\t;'): continue
    cur[1] += 1
    if 'scratch_' in l: cur[2] += 1
    if 'v_accvgpr' in l: cur[3] += 1
    if '_dpp' in l: cur[4] += 1
    if re.match(r'\tv_(fma|mul|add|max|min)_f64', l): cur[5] += 1
    if l.startswith('\tds_'): cur[6] += 1
    if l.startswith('\tv_cndmask'): cur[7] += 1
print('%-60s %7s %7s %7s %6s %6s %5s %6s' % ('block', 'instr', 'scratch', 'accvgpr', 'dpp', 'f64', 'ds', 'cndmsk'))
tot = {}
for b in blocks:
    if 'entry' in b[0]: print('==', b[0])
    if b[1] >= minb: print('%-60s %7d %7d %7d %6d %6d %5d %6d' % tuple([b[0][:60]] + b[1:]))
