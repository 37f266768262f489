cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt | cut -c1-400
