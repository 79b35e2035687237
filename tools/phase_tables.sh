#!/bin/bash
# Dev tool (GPU box): per-phase tick tables of the set-up of C3 / C4 (16-lane-row kernels) and C5 (workgroup kernel) from the
# instrumented builds of tools/prof_build.sh, plus the product build's phase scan.  usage: bash tools/phase_tables.sh <outdir>
set -e
OUT=${1:-gpurun_out/phase}
mkdir -p $OUT
for c in 3 4; do
  LQMPC_LIB=build_prof/liblqmpc_prof.so python3 tools/r16_prof.py $c > $OUT/r16_prof_c$c.txt 2>&1
done
LQMPC_LIB=build_prof/liblqmpc_wgprof.so python3 tools/wg_prof.py > $OUT/wg_prof_c5.txt 2>&1
for c in 3 4; do python3 tools/phase_scan.py $c > $OUT/phase_scan_c$c.txt 2>&1; done
python3 tools/phase_scan.py 5 8192 > $OUT/phase_scan_c5_8192.txt 2>&1
tail -n 8 $OUT/*.txt
