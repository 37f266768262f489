# round 4, first GPU trip: the new tests first (fused GL chain, stream order / hipGraph capture, 2-rank bench), then the whole parity suite, then the bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests/test_gl_fused.py tests/test_stream_order.py tests/test_gl_storage.py tests/test_gl_reference.py -q -m gpu -x > $O/pytest_new.txt 2>&1
tail -25 $O/pytest_new.txt | cut -c1-300
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -40
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc $?
tail -5 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench_line.json'))
print('bench', d['value'], d['roofline']['frac'], 'sustained', d.get('sustained',{}).get('roofline_frac'), 'strict', d['strict_log']['roofline_frac'], 'chain', d['smooth_chain']['roofline_frac'], 'r16', d['r16_texels']['roofline_frac'])
for k,v in d['configs'].items():
    print(k, round(v['avg_kernel_ms'],4), round(v['roofline_frac'],4), {kk: (round(vv['avg_kernel_ms'],4), round(vv['roofline_frac'],4)) for kk,vv in v.items() if isinstance(vv,dict) and 'roofline_frac' in vv})
print(d['cpu_baseline'].get('configs[0]'))
PY
