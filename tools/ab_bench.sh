#!/bin/bash
# tools/ab_bench.sh <tag> <variant-name> [cmd...] -- run on the GPU box: the product library and an A/B build of it
# (python -m glava_amd.build --variant NAME FLAGS..., glava_amd/csrc/libglvspectrum_NAME.so) through the same
# benchmarks, alternating, in one call.  Default benchmarks: the chain and the configuration table rows.
cd $GRAFT_REPO_ROOT
TAG=$1; VAR=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/r02/ab_$TAG.txt
mkdir -p $(dirname $O); : > $O
ALT=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum_$VAR.so
CMDS=("$@")
if [ ${#CMDS[@]} -eq 0 ]; then CMDS=("python tools/chain_bench.py" "python tools/gravity_bench.py"); fi
for rep in 1 2; do
  for c in "${CMDS[@]}"; do
    echo "== product: $c" >> $O; $c 2>/dev/null >> $O
    echo "== $VAR: $c" >> $O; GLV_SPECTRUM_LIB=$ALT $c 2>/dev/null >> $O
  done
done
cat $O
