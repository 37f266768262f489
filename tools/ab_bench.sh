#!/bin/bash
# tools/ab_bench.sh <tag> <variant[,variant...]> [cmd...] -- run on the GPU box: the product library and A/B builds of
# it (glava_amd.build.build_variant(NAME, FLAGS) -> glava_amd/csrc/libglvspectrum_NAME.so) through the same benchmarks,
# alternating, twice, in one call.  Default benchmarks: the bars table and the chain.
cd $GRAFT_REPO_ROOT
TAG=$1; IFS=',' read -ra VARS <<< "$2"; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/r02/ab_$TAG.txt
mkdir -p $(dirname $O); : > $O
CMDS=("$@")
if [ ${#CMDS[@]} -eq 0 ]; then CMDS=("python tools/bars_bench.py" "python tools/chain_bench.py"); fi
for rep in 1 2; do
  for c in "${CMDS[@]}"; do
    echo "== product: $c" >> $O; $c 2>/dev/null >> $O
    for v in "${VARS[@]}"; do
      echo "== $v: $c" >> $O; GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum_$v.so $c 2>/dev/null >> $O
    done
  done
done
cat $O
