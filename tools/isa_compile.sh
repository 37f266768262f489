#!/bin/bash
# tools/isa_compile.sh <log2(nn)> '<variant list>' [extra -D flags...]  -- compile one glv_tune.hip variant set with
# -save-temps into /tmp/isa/<tag>/ and print the register / scratch / instruction-mix summary (tools/isa_stats.py).
set -e
K=$1; V=$2; shift 2
TAG=n${K}_$(echo "$V $*" | md5sum | cut -c1-6)
D=/tmp/isa/$TAG
mkdir -p $D
cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DGLV_TUNE_LOG_NN=$K -DGLV_TUNE_NO_LOG0 \
    "-DGLV_TUNE_VARIANTS=$V" "$@" -save-temps -c /root/repo/glava_amd/csrc/glv_tune.hip -o $D/t.o
python /root/repo/tools/isa_stats.py $D/glv_tune-hip-amdgcn-amd-amdhsa-gfx950.s
echo "asm: $D/glv_tune-hip-amdgcn-amd-amdhsa-gfx950.s"
