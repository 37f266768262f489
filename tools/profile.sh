#!/bin/bash
# tools/profile.sh -- rocprofv3 evidence for bench.py's dominant kernel (run on the GPU box).
#   1. --kernel-trace --stats : per-kernel average duration (must agree with bench.py's HIP events)
#   2. separate --pmc passes  : SQ busy/wait split, LDS conflicts, HBM FETCH_SIZE / WRITE_SIZE
# PMC passes never combine with sys/hip/hsa tracing (gpurun refuses that; it crashes nodes).
# Usage: tools/profile.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
PMCBENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic $*"
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_COUNT" \
  "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -o bench -- $PMCBENCH > /dev/null 2> "$OUT/pmc$i.err"
done
python "$ROOT/tools/prof_summary.py" "$OUT" ${TRAFFIC_ARGS:-} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
