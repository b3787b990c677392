#!/bin/bash
# memory-latency counters for the guided-filter variants: bash profiles/pmc_gf2.sh <tag> [gf_bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sqgf2_$TAG
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/profiles/gf_bench.py "$@" > $OUT.a.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "guided" not in k: continue
        k = k.split("uwie::(anonymous namespace)::", 1)[-1].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, {c: "%.4g" % x for c, x in v.items()})
        if v.get("SQ_INSTS_VMEM"): print("   vmem latency ~ %.0f cycles (level/insts), lds ~ %.0f" % (v["SQ_INST_LEVEL_VMEM"] / v["SQ_INSTS_VMEM"], v["SQ_INST_LEVEL_LDS"] / max(v["SQ_INSTS_LDS"], 1)))
PY
