#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/prof_traffic.sh <tag> [bench args...]  -- kernel trace + FETCH_SIZE / WRITE_SIZE passes only
set -o pipefail
TAG=$1; shift
# one process only: rocprofv3's preloaded runtime has initialised the GPU, and `--gpus N` makes bench.py a launcher chain (forbidden exec)
case " $* " in *" --gpus "[2-9]*) echo "prof: profile one rank (--gpus 1)"; exit 2;; esac
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-extras $@"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$P -- python3 bench.py $ARGS > $OUT/pmc_$P.log 2>&1 || echo "pmc pass failed: $P"
done
python3 tools/prof_summary.py $OUT 2>/dev/null | grep -E "lqmpc.*(FETCH_SIZE|WRITE_SIZE)" | cut -c1-150
