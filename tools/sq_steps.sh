#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/sq_steps.sh <tag> [config]  -- vector instructions of the rollout kernel for T = 1..10 (default mix)
set -o pipefail
TAG=$1; CFG=${2:-3}
OUT=gpurun_out/sqs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cat > $OUT/run.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, _lib
if os.environ.get('LQMPC_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['LQMPC_LIB'])
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
b = synth.make_batch(int(sys.argv[1]))
nx, nu, N, Bsz = b['A'].shape[0], b['B'].shape[1], b['N'], b['Bsz']
dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev); dx0 = torch.from_numpy(b['x0']).to(dev)
dJ = torch.empty(Bsz, dtype=torch.float64, device=dev); dit = torch.empty(Bsz, dtype=torch.int32, device=dev)
for T in range(1, 11):
    for _ in range(3):
        s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0, b['A_true'], b['B_true'], dJ, diters=dit)
    torch.cuda.synchronize()
    print(T, dit.double().sum().item() / Bsz, flush=True)
PY
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $OUT/run.py $CFG > $OUT/p1.log 2>&1 || echo "pass failed"
cat $OUT/p1.log | grep -v amdgpu
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/p1/*/*counter_collection.csv'):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        k = r['Kernel_Name']
        if 'lqmpc_r16_kernel' in k or 'lqmpc_r64_kernel' in k: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(acc.items()):
    per = len(v) // 10
    m = [sum(v[g * per:(g + 1) * per]) / per for g in range(10)]
    print(c, ' '.join('%.4g' % x for x in m))
    print('   per step:', ' '.join('%.4g' % (m[g] - m[g - 1]) for g in range(1, 10)))
PY
