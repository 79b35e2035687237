#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/sq_phase.sh <tag> [config]  -- instruction counts of the rollout kernel per (mix, T) of tools/phase_scan.py
set -o pipefail
TAG=$1; CFG=${2:-3}
OUT=gpurun_out/sqp_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 tools/phase_scan.py $CFG > $OUT/p1.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/p1/*/*counter_collection.csv'):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        k = r['Kernel_Name']
        if 'lqmpc_r16_kernel' in k or 'lqmpc_r64_kernel' in k:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
names = ['default T=1', 'default T=2', 'default T=8', 'default T=30', 'tiny T=1', 'tiny T=2', 'tiny T=8', 'tiny T=30']
for c, v in sorted(acc.items()):
    per = len(v) // 8
    print(c, ' '.join('%s: %.4g' % (names[g], sum(v[g * per:(g + 1) * per]) / per) for g in range(8)))
PY
