# round 5: the GL-default chain with the pre-smoothing pass -- parity subset, timing, kernel trace + PMC (SQ busy / wait, LDS conflicts, MFMA busy)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
timeout 600 python -m pytest tests/test_gl_fused.py -q -m gpu -x > $O/pytest_sm.txt 2>&1
tail -4 $O/pytest_sm.txt | cut -c1-300
timeout 300 python tools/cfg_run.py gl_sm 100 2>&1 | tail -1 | tee $O/cfg_i8.txt
timeout 300 python tools/sm_overlap.py 16384 60 2>&1 | tail -2 | tee $O/overlap.txt
GLV_PMC_EXTRA="SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" timeout 900 bash tools/profile_cmd.sh r05_gl_sm python $GRAFT_REPO_ROOT/tools/cfg_run.py gl_sm 100 > $O/prof_gl_sm.txt 2>&1
grep -E "glv_|traffic|wall" gpurun_out/prof_r05_gl_sm/summary.txt | cut -c1-400 | head -30
