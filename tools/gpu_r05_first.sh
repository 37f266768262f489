# round 5, first GPU call: the integer (i8 matrix-core) pre-smoothing pass -- parity, timing against the f32 pass, kernel trace, the two-stream overlap
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_gl_fused.py tests/test_gl_reference.py tests/test_stream_order.py tests/test_glsl_twins.py -q -m gpu -x > $O/pytest_first.txt 2>&1
tail -15 $O/pytest_first.txt | cut -c1-400
for c in gl_sm gl_default; do timeout 300 python tools/cfg_run.py $c 100 2>&1 | tail -1; done | tee $O/cfg_i8.txt
GLV_NO_BARS_I8=1 timeout 300 python tools/cfg_run.py gl_sm 100 2>&1 | tail -1 | tee $O/cfg_f32.txt
timeout 300 python tools/sm_overlap.py 16384 60 2>&1 | tail -3 | tee $O/overlap.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gl_sm -o bench -- python $GRAFT_REPO_ROOT/tools/cfg_run.py gl_sm 100 > $O/prof_gl_sm.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_gl_sm -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-250
