cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu.txt | head -20
timeout 600 python tools/configs_bench.py > $O/configs.txt 2>&1; grep -E "config\[2|N=16384|N= 8192 x 32768" $O/configs.txt
timeout 300 python tools/bars_bench.py > $O/bars.txt 2>&1; tail -12 $O/bars.txt
