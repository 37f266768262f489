set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/r02_sweep.py --run > gpurun_out/r02/sweep_stdout.txt 2>&1
cp gpurun_out/r02/sweep.txt gpurun_out/r02/sweep2.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r02/pytest_gpu.txt
python bench.py > gpurun_out/r02/bench1.json 2> gpurun_out/r02/bench1.err
cat gpurun_out/r02/bench1.json
