#!/bin/bash
# round 4: the whole GPU suite, then the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -20
python tools/sm_bench.py 4096 2048 1024 2>&1 | grep -v amdgpu.ids | tee $O/sm_bench.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_line_3.json 2> $O/bench3.err; echo bench rc $?
python - <<PY
import json
d=json.load(open('$O/bench_line_3.json'))
print(round(d['ms_per_step'],4), round(d['roofline']['frac'],4))
g=d['configs']['gl_default']
print({k:(round(v['avg_kernel_ms'],4), round(v['value']/1e6,3), round(v['roofline_frac'],4), v.get('launches_per_step')) for k,v in g.items() if isinstance(v,dict)}, round(g['avg_kernel_ms'],4), round(g['roofline_frac'],4))
PY
