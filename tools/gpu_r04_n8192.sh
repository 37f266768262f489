#!/bin/bash
# round 4: N = 8192 after the padding period of its pass-0 exchange went from 16 to 32 elements (LDS bank conflicts)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "8192 or sizes or raw" 2>&1 | grep -E "passed|failed" | tail -2
python tools/grid_ab.py 8192 256 2>&1 | grep -v amdgpu
bash tools/profile.sh r04_n8192 --n 8192 --streams 32768 --no-alt --no-configs --sustained-s 0 > /dev/null 2>&1
grep -E "glv_frame_kernel<12|LDS_BANK|WAIT_INST_LDS|INSTS_LDS" gpurun_out/prof_r04_n8192/summary.txt | cut -c1-200
