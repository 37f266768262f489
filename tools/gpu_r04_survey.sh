cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
python tools/variant_survey.py --out $O/variants.txt --wisdom $O/wisdom_mi355x.txt > /dev/null 2> $O/variants.err; cat $O/variants.txt | cut -c1-150 | grep -v autotune
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc $?
python - <<PY
import json
d=json.load(open('$O/bench_line.json'))
g=d['configs']['gl_default']
print({k:(round(v['avg_kernel_ms'],4), round(v['value']/1e6,3), round(v['roofline_frac'],4)) for k,v in g.items() if isinstance(v,dict)}, round(g['avg_kernel_ms'],4), round(g['roofline_frac'],4))
PY
