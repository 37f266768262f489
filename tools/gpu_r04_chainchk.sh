#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python tools/cfg_run.py chain 150 2>&1 | grep -v amdgpu
for st in 20 150; do
python bench.py --no-cpu-baseline --steps $st > gpurun_out/r04/bench_line_4.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_line_4.json"))
print("bench steps $st: chain", d["smooth_chain"]["avg_kernel_ms"], d["smooth_chain"]["ms_per_step"], "headline", d["roofline"]["avg_kernel_ms"], d["ms_per_step"], "sustained", d["sustained"]["avg_kernel_ms"], "traffic", d["roofline"]["traffic"], "r16", d["r16_texels"]["avg_kernel_ms"], "gl", d["configs"]["gl_default"]["avg_kernel_ms"])
PY
done
python tools/cfg_run.py chain 150 2>&1 | grep -v amdgpu
python tools/cfg_run.py gl_default 150 2>&1 | grep -v amdgpu
