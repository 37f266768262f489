cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
timeout 1500 python -m pytest tests/test_gl_fused.py tests/test_gl_reference.py tests/test_handle_audio.py tests/test_gpu_parity.py -q -m gpu > $O/pytest_c.txt 2>&1
tail -25 $O/pytest_c.txt | cut -c1-600
