set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r02/pytest_gpu.txt
TRAFFIC_ARGS="--traffic-json $GRAFT_REPO_ROOT/gpurun_out/r02/hbm_traffic.json --n 4096 --streams 65536 --ops fft --kernel ELi0ELi0E" bash tools/profile.sh r02 --no-alt > /dev/null 2>&1
bash tools/profile.sh r02_n8192 --n 8192 --streams 32768 --no-alt > /dev/null 2>&1
bash tools/profile.sh r02_n16384 --n 16384 --streams 16384 --no-alt > /dev/null 2>&1
ls gpurun_out/prof_r02*
