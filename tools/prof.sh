#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/prof.sh <tag> [bench args...]
# kernel-trace stats, then separate PMC passes (never combined with trace domains other than kernel-trace).
set -o pipefail
TAG=$1; shift
# one process only: rocprofv3's preloaded runtime has initialised the GPU, and `--gpus N` makes bench.py a launcher chain (forbidden exec)
case " $* " in *" --gpus "[2-9]*) echo "prof: profile one rank (--gpus 1)"; exit 2;; esac
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_IFETCH SQ_INSTS_SCRATCH SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
  N=$(echo $P | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- python3 bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || echo "pmc pass failed: $P"
done
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
