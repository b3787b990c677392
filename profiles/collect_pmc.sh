#!/bin/bash
# Collects rocprofv3 PMC passes for bench.py on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect_pmc.sh <tag> [bench args...]
# One pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit together; no trace domains besides kernel-trace).
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras "${BENCH_ARGS[@]}" > $OUT.$name.log 2>&1
}
BENCH_ARGS=("$@")
mkdir -p $OUT
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run fetch FETCH_SIZE
run write WRITE_SIZE
run req TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum
ls -R $OUT | head -30
