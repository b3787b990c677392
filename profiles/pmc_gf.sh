#!/bin/bash
# SQ issue/stall counters for the guided-filter variants of profiles/gf_bench.py (run through gpurun from the repo root):
#   bash profiles/pmc_gf.sh <tag> [gf_bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sqgf_$TAG
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/profiles/gf_bench.py "$@" > $OUT.a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/profiles/gf_bench.py "$@" > $OUT.b.log 2>&1
python3 $GRAFT_REPO_ROOT/profiles/summarize_sq.py $OUT guided > $OUT.summary.txt
cat $OUT.summary.txt
