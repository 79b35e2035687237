#!/bin/bash
# Dev tool: build the instrumented 16-lane-row kernel (clock64 accounting, -DLQMPC_R16_PROF) into build_prof/liblqmpc_prof.so.
# usage (from anywhere): bash tools/prof_build.sh [PROFBLK]; run with LQMPC_LIB=build_prof/liblqmpc_prof.so python tools/r16_prof.py 3 [hard]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build_prof"
make -C "$ROOT/lq_mpc_amd/csrc" -j8 > /dev/null
for f in lqmpc_api lqmpc_bounds lqmpc_spec lqmpc_generic lqmpc_wg lqmpc_r16_lat lqmpc_jit; do cp "$ROOT/lq_mpc_amd/csrc/$f.o" "$ROOT/build_prof/"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DLQMPC_R16_PROF -DPROFBLK=${1:-100} \
    -c "$ROOT/lq_mpc_amd/csrc/lqmpc_r16.hip" -o "$ROOT/build_prof/lqmpc_r16.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_prof/liblqmpc_prof.so" "$ROOT"/build_prof/*.o -ldl
echo built "$ROOT/build_prof/liblqmpc_prof.so"
# the workgroup kernel's instrumented build (block 0's phases): build_prof/liblqmpc_wgprof.so
mkdir -p "$ROOT/build_prof/wg"
for f in lqmpc_api lqmpc_bounds lqmpc_spec lqmpc_generic lqmpc_r16 lqmpc_r16_lat lqmpc_jit; do cp "$ROOT/lq_mpc_amd/csrc/$f.o" "$ROOT/build_prof/wg/"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DLQMPC_WG_PROF -c "$ROOT/lq_mpc_amd/csrc/lqmpc_wg.hip" -o "$ROOT/build_prof/wg/lqmpc_wg.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_prof/liblqmpc_wgprof.so" "$ROOT"/build_prof/wg/*.o -ldl
echo built "$ROOT/build_prof/liblqmpc_wgprof.so"
