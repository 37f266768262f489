cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -40
python tools/gl_bench.py > $O/gl_bench.txt 2>/dev/null; cat $O/gl_bench.txt
python tools/gl_bench.py 1024 >> $O/gl_bench.txt 2>/dev/null; python tools/gl_bench.py 16384 >> $O/gl_bench.txt 2>/dev/null; tail -8 $O/gl_bench.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc $?
python - <<PY
import json
d=json.load(open('$O/bench_line.json'))
print('bench', d['value'], d['roofline']['frac'], 'sustained', d.get('sustained',{}).get('roofline_frac'), 'strict', d['strict_log']['roofline_frac'], 'chain', d['smooth_chain']['roofline_frac'], 'r16', d['r16_texels']['roofline_frac'])
for k,v in d['configs'].items():
    print(k, round(v['avg_kernel_ms'],4), round(v['roofline_frac'],4), {kk: (round(vv['avg_kernel_ms'],4), round(vv['roofline_frac'],4)) for kk,vv in v.items() if isinstance(vv,dict) and 'roofline_frac' in vv})
PY
