# Round-4 evidence run (one gpurun call): GPU parity suite, bench line, rocprofv3 stats + PMC of the headline kernel and of the
# configurations VERDICT r3 asked to re-profile warm (>= 100 calls after spin-up): GL-default (av / bars), configs[2], the chain,
# N=1024 bars, the ring update, N=8192 / N=16384 stateless.  Everything lands in gpurun_out/r04/ and gpurun_out/prof_r04*/.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
python bench.py > $O/bench_line.json 2> $O/bench.err
TRAFFIC_ARGS="--traffic-json $O/hbm_traffic.json --n 4096 --streams 65536 --ops fft --kernel glv_frame_kernel<11,~0,~1,~2,~1,~1,~true,~2,~1,~1,~4,~0,~0>" bash tools/profile.sh r04 --no-alt --no-configs --sustained-s 0 > $O/prof_r04.txt 2>&1
bash tools/profile.sh r04_n8192 --n 8192 --streams 32768 --no-alt --no-configs --sustained-s 0 > /dev/null 2>&1
bash tools/profile.sh r04_n16384 --n 16384 --streams 16384 --no-alt --no-configs --sustained-s 0 > /dev/null 2>&1
for c in gl_default gl_bars gl_sm configs2 chain n1024bars ring; do
  bash tools/profile_cmd.sh r04_$c python $GRAFT_REPO_ROOT/tools/cfg_run.py $c 150 > $O/prof_$c.txt 2>&1
done
python tools/power_probe.py --seconds 5 > $O/power.txt 2>/dev/null
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > $O/device.txt; nproc >> $O/device.txt
tail -12 $O/prof_r04.txt | cut -c1-200
for c in gl_default gl_bars configs2 chain n1024bars ring; do echo "== $c"; grep -E "glv_frame_kernel|traffic|wall ms" gpurun_out/prof_r04_$c/summary.txt gpurun_out/prof_r04_$c/cmd_stats.txt | cut -c1-260 | head -8; done
cat $O/power.txt
