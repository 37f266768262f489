#!/bin/bash
# round 4: the matrix-core pre-smoothing kernel: parity, then timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gl_fused.py tests/test_gl_reference.py tests/test_glsl_twins.py tests/test_gl_storage.py tests/test_stream_order.py -q -m gpu -x > $O/pytest_rows2.txt 2>&1
tail -15 $O/pytest_rows2.txt | cut -c1-250
for n in 4096 2048 1024 512; do tools/bin/rows_bench $n; done 2>&1 | grep -v amdgpu.ids | tee $O/rows_bench.txt
python tools/sm_bench.py 4096 2048 2>&1 | grep -v amdgpu.ids | tee $O/sm_bench2.txt
