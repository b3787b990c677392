#!/bin/bash
# SQ issue / stall counters of the exact-order guided filter's kernels (profiles/time_exact.py under rocprofv3, two counter passes):
#   bash profiles/pmc_sq_exact.sh <tag> [H W B]          -> gpurun_out/<tag>_exact_sq_summary.txt
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/profiles/time_exact.py "$@" > $OUT.a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python $GRAFT_REPO_ROOT/profiles/time_exact.py "$@" > $OUT.b.log 2>&1
python3 $GRAFT_REPO_ROOT/profiles/summarize_sq.py $OUT k_box > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_exact_sq_summary.txt 2>&1
rm -rf $OUT $OUT.a.log $OUT.b.log
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_exact_sq_summary.txt
