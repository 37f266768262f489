cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests/test_gl_fused.py tests/test_gl_reference.py tests/test_glsl_twins.py tests/test_gpu_parity.py -q -m gpu > $O/pytest_rows.txt 2>&1
tail -3 $O/pytest_rows.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest_rows.txt | head
timeout 900 python bench.py --no-cpu-baseline > $O/bench_line_rows.json 2> $O/bench.err; echo bench rc $?
python - <<PY
import json
d=json.load(open('$O/bench_line_rows.json'))
g=d['configs']['gl_default']
print({k:(round(v['avg_kernel_ms'],4), round(v['value']/1e6,3), round(v['roofline_frac'],4), v.get('launches_per_step')) for k,v in g.items() if isinstance(v,dict)}, round(g['avg_kernel_ms'],4), round(g['roofline_frac'],4))
PY
