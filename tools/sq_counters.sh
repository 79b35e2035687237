#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/sq_counters.sh <tag> <config>  -- SQ issue / wait counters of the rollout kernels (own passes, kernel trace only)
set -o pipefail
TAG=$1; CFG=${2:-3}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
         "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
         "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 tools/mix_time.py $CFG > $OUT/p$i.log 2>&1 || echo "pass failed: $P"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'lqmpc' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        h = len(v) // 2      # first half of the dispatches = default mix, second half = hard mix (mix_time.py order)
        print(f"   {c:32s} default {sum(v[:h]) / max(h, 1):16.1f}   hard {sum(v[h:]) / max(len(v) - h, 1):16.1f}   n={len(v)}")
PY
