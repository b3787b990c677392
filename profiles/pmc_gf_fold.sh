#!/bin/bash
# FETCH_SIZE per k_guided_split dispatch (gf_bench.py runs the XCD-folded launch order first, then the plain one):
#   bash profiles/pmc_gf_fold.sh <tag> [gf_bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/foldgf_$TAG
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $GRAFT_REPO_ROOT/profiles/gf_bench.py "$@" > $OUT.f.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_guided_split" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    print("k_guided_split FETCH_SIZE x2 per dispatch, GB:", ["%.3f" % (2 * float(r["Counter_Value"]) * 1024 / 1e9) for r in rows])
PY
