# run the current tools/r02_sweep.py batch on the GPU box; $1 = tag of the output copy
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/r02_sweep.py --run > gpurun_out/r02/sweep_stdout.txt 2>&1
cp gpurun_out/r02/sweep.txt gpurun_out/r02/sweep_$1.txt
grep -v amdgpu.ids gpurun_out/r02/sweep_$1.txt | cut -c1-190
