#!/bin/bash
# round 4: the lane-per-row pre-smoothing kernel with parts removed (GLV_EXP_ROWS_*: wrong results, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for v in "" rows_nofill rows_noflush rows_nocompute rows_computeonly ""; do
  if [ -n "$v" ]; then export GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum_$v.so; else unset GLV_SPECTRUM_LIB; fi
  echo "== ${v:-product}"
  python tools/sm_bench.py 4096 2048 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/rows_parts.txt 2>&1
cat gpurun_out/rows_parts.txt
