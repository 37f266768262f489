# Round-2 evidence run (one gpurun call): GPU parity suite, bench line, rocprofv3 stats + PMC of bench.py's kernel
# (N=4096) and of the two larger sizes, the configuration tables.  Everything lands in gpurun_out/r02*/.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
python bench.py > $O/bench_line.json 2> $O/bench.err
TRAFFIC_ARGS="--traffic-json $O/hbm_traffic.json --n 4096 --streams 65536 --ops fft --kernel glv_frame_kernel<11,~0,~1,~2,~1,~1,~true,~2,~1,~1,~4,~0,~0>" bash tools/profile.sh r02 --no-alt > /dev/null 2>&1
bash tools/profile.sh r02_n8192 --n 8192 --streams 32768 --no-alt > /dev/null 2>&1
bash tools/profile.sh r02_n16384 --n 16384 --streams 16384 --no-alt > /dev/null 2>&1
python tools/configs_bench.py --out $O/configs.txt > /dev/null 2> $O/configs.err
python tools/chain_bench.py > $O/chain.txt 2>/dev/null
GLV_UNFUSED_BARS=1 python tools/chain_bench.py 2>/dev/null | tail -1 >> $O/chain.txt
python tools/gravity_bench.py > $O/gravity.txt 2>/dev/null
python tools/bars_bench.py > $O/bars.txt 2>/dev/null
python tools/smooth_bench.py > $O/smooth.txt 2>/dev/null
python tools/inputs_bench.py > $O/inputs.txt 2>/dev/null
tools/bin/membench2 > $O/membench2.txt 2>&1
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > $O/device.txt; nproc >> $O/device.txt
cat $O/configs.txt $O/chain.txt
python -c "
import json; d=json.load(open('$O/bench_line.json')); print('bench', d['value'], d['roofline']['frac'], 'strict', d['strict_log']['roofline_frac'], 'chain', d['smooth_chain']['roofline_frac'], 'r16', d['r16_texels']['value'], d['r16_texels']['roofline_frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
