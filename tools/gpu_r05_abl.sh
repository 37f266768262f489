cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
for v in "" _nostage _nostore _nostage_nostore _noepi _nomfma _nowload _noepi_nomfma _noepi_nomfma_nowload; do
  echo "== rows_i8_bench$v"; timeout 120 tools/bin/rows_i8_bench$v 4096 32768 20 2>&1 | tail -1
done | tee $O/i8_ablation.txt
