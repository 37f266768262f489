# round 3, first GPU trip: parity suite + the bench line (with the new `configs` key)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu.txt | head -20
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc $?
tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench_line.json'))
print('bench', d['value'], d['roofline']['frac'], 'strict', d['strict_log']['roofline_frac'], 'chain', d['smooth_chain']['roofline_frac'], 'r16', d['r16_texels']['roofline_frac'])
for k,v in d['configs'].items():
    print(k, round(v['avg_kernel_ms'],4), round(v['roofline_frac'],4), {kk: round(vv['roofline_frac'],4) for kk,vv in v.items() if isinstance(vv,dict)})
    if 'classes' in v:
        for c in v['classes']: print('   ', c['n'], round(c['avg_kernel_ms'],4), round(c['roofline_frac'],4))
PY
python tools/variant_survey.py --out $O/variants.txt --wisdom $O/wisdom_mi355x.txt > /dev/null 2> $O/variants.err; cat $O/variants.txt | cut -c1-200
