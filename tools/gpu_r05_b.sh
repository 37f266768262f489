cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
for v in "" _nostage _nostore; do echo "== rows_i8_bench$v"; timeout 120 tools/bin/rows_i8_bench$v 4096 32768 20 2>&1 | tail -1; done | tee $O/i8_ablation2.txt
timeout 900 python -m pytest tests/test_gl_fused.py tests/test_stream_order.py -q -m gpu -x > $O/pytest_sm.txt 2>&1
tail -6 $O/pytest_sm.txt | cut -c1-400
timeout 300 python tools/cfg_run.py gl_sm 100 2>&1 | tail -1 | tee $O/cfg_i8.txt
GLV_SM_NO_SPLIT=1 timeout 300 python tools/cfg_run.py gl_sm 100 2>&1 | tail -1 | tee -a $O/cfg_i8.txt
