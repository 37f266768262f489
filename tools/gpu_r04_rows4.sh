#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GLV_PMC_EXTRA="SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT"
tools/profile_cmd.sh rows_v3 $GRAFT_REPO_ROOT/tools/bin/rows_bench 4096 32768 288 5 2>&1 | grep -v "^\[" | tail -32
tools/profile_cmd.sh rows_v3_co $GRAFT_REPO_ROOT/tools/bin/rows_bench_computeonly 4096 32768 288 5 2>&1 | tail -32
