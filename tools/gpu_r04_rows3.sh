#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for v in "" _nofill _noflush _nocompute _computeonly ""; do
  echo "== rows_bench$v"; tools/bin/rows_bench$v 4096; tools/bin/rows_bench$v 2048
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/rows_parts2.txt
cat gpurun_out/rows_parts2.txt
