#!/bin/bash
# tools/profile_cmd.sh <tag> <command...> -- like tools/profile.sh for an arbitrary command (run on the GPU box):
# rocprofv3 --kernel-trace --stats, then separate --pmc passes (HBM FETCH_SIZE / WRITE_SIZE, SQ busy/wait), summary by
# tools/prof_summary.py -> gpurun_out/prof_<tag>/summary.txt.  PMC passes never combine with sys/hip/hsa tracing.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- "$@" > "$OUT/cmd_stats.txt" 2> "$OUT/stats.err"
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  ${GLV_PMC_EXTRA:+"$GLV_PMC_EXTRA"} \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -o bench -- "$@" > /dev/null 2> "$OUT/pmc$i.err"
done
python "$ROOT/tools/prof_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
