#!/bin/bash
# SQ issue/stall counters for bench.py (run through gpurun from the repo root): bash profiles/pmc_sq.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT.a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT.b.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "ab":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("uwie::", "").split("(")[0][:60]
            if "${KFILTER:-guided}" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            print(k, {c: "%.4g" % x for c, x in v.items()})
PY
