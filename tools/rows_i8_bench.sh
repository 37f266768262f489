#!/bin/bash
# build tools/bin/rows_i8_bench[_<name>] (name = "": the kernel as shipped; otherwise with the given -D switches)
cd "$(dirname "$0")/.." || exit 1
name=${1:+_$1}; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -DGLV_TUNE_BUILD "$@" tools/rows_i8_bench.hip -o tools/bin/rows_i8_bench$name
