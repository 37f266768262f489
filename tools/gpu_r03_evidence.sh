# Round-3 evidence run (one gpurun call): GPU parity suite, bench line, rocprofv3 stats + PMC of bench.py's kernel (N=4096) and
# of the two larger sizes, the configuration table.  Everything lands in gpurun_out/r03*/ and gpurun_out/prof_r03*/.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
python bench.py > $O/bench_line.json 2> $O/bench.err
TRAFFIC_ARGS="--traffic-json $O/hbm_traffic.json --n 4096 --streams 65536 --ops fft --kernel glv_frame_kernel<11,~0,~1,~2,~1,~1,~true,~2,~1,~1,~4,~0,~0>" bash tools/profile.sh r03 --no-alt --no-configs > $O/prof_r03.txt 2>&1
bash tools/profile.sh r03_n8192 --n 8192 --streams 32768 --no-alt --no-configs > /dev/null 2>&1
bash tools/profile.sh r03_n16384 --n 16384 --streams 16384 --no-alt --no-configs > /dev/null 2>&1
python tools/configs_bench.py --out $O/configs.txt > /dev/null 2> $O/configs.err
python tools/bars_bench.py > $O/bars.txt 2>/dev/null
python tools/bars_unfused_bench.py > $O/bars_unfused.txt 2>/dev/null
python tools/bars_probe.py 16384 8192 > $O/bars_probe.txt 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_bars/stats -o bars -- python $GRAFT_REPO_ROOT/tools/bars_bench.py > /dev/null 2>&1)
python tools/prof_summary.py gpurun_out/prof_r03_bars > $O/rocprofv3_bars_summary.txt 2>/dev/null || ls -R gpurun_out/prof_r03_bars | head -5
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > $O/device.txt; nproc >> $O/device.txt
cat $O/configs.txt | cut -c1-220
tail -30 $O/prof_r03.txt | cut -c1-200
