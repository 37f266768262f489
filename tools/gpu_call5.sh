cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/r02_sweep.py --run > gpurun_out/r02/sweep_stdout.txt 2>&1
cp gpurun_out/r02/sweep.txt gpurun_out/r02/sweep3.txt
grep -v amdgpu.ids gpurun_out/r02/sweep3.txt | cut -c1-175
