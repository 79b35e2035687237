#!/bin/bash
# GPU box: the round's record -- bench lines of C2..C5, rocprofv3 kernel traces + PMC passes of C3 / C4 / C5, the world-size-1 RCCL runs.
# usage: bash tools/round_profiles.sh <round tag, e.g. r03>
set -o pipefail
R=${1:-r03}
O=gpurun_out/${R}_final
mkdir -p $O
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err
for c in 2 4 5; do python bench.py --config $c --steps 10 --warmup 2 > $O/bench_C$c.json 2> $O/bench_C$c.err; done
for c in 3 4 5; do bash tools/prof.sh ${R}_c$c --config $c > $O/prof_c$c.log 2>&1; done
python bench.py --force-dist --config 4 --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_C4_nccl_world1.json 2> $O/bench_C4_nccl_world1.err
python bench.py --force-dist --gather per-step --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_C3_nccl_world1_per_step.json 2> $O/bench_C3_nccl_world1_per_step.err
ls -la $O
