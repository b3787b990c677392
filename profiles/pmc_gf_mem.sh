#!/bin/bash
# HBM-side traffic counters for the guided-filter variants: bash profiles/pmc_gf_mem.sh <tag> [gf_bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/memgf_$TAG
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
run() { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/profiles/gf_bench.py $ARGS > $OUT.$n.log 2>&1; }
ARGS="$@"
run f FETCH_SIZE
run w WRITE_SIZE
run h TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run d TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "guided" not in k: continue
        k = k.split("uwie::(anonymous namespace)::", 1)[-1].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if "/f/" in f and r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); calls[k] += 1
for k, v in acc.items():
    n = max(calls[k], 1)
    print(k, "calls", n, {c: "%.4g" % (x / n) for c, x in sorted(v.items())})
PY
