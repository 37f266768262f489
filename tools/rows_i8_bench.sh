#!/bin/bash
# build tools/bin/rows_i8_bench[_<name>] (name = "": the kernel as shipped; otherwise with the given -D switches);
# name = x16: the v_mfma_i32_16x16x64_i8 prototype tools/rows_i8x16_bench.hip -> tools/bin/rows_i8x16_bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -DGLV_TUNE_BUILD"
if [ "$1" = "x16" ]; then shift; exec /opt/rocm/bin/hipcc $FLAGS "$@" tools/rows_i8x16_bench.hip -o tools/bin/rows_i8x16_bench; fi
name=${1:+_$1}; shift
/opt/rocm/bin/hipcc $FLAGS "$@" tools/rows_i8_bench.hip -o tools/bin/rows_i8_bench$name
