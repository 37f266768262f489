set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/r02_sweep.py --run > gpurun_out/r02/sweep_stdout.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "r16 or raw_bit_exact or gravity_average or fifo_ring or bars" > gpurun_out/r02/pytest_r16.txt 2>&1
tail -5 gpurun_out/r02/pytest_r16.txt
