#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
for n in 4096 2048 1024 512; do tools/bin/rows_bench $n; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/rows_bench.txt
