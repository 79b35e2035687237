#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/trace_one.sh <tag> <python script + args...>  -- kernel trace; prints the average duration per kernel
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/tr_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1 || echo "run failed"
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'lqmpc' in r['Name']: print('%-64s calls %5s  avg %9.2f us' % (r['Name'].replace('lqmpc::', '')[:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
