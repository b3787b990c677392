#!/bin/bash
# Kernel timeline of one 1080p batch-1 step (BASELINE.json configs[1]): bash profiles/timeline_1080p.sh <tag>
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -- python3 $R/bench.py --height 1080 --width 1920 --batch 1 --no-extras --no-cpu-baseline --steps 20 > $R/gpurun_out/tl_$TAG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/tl_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_chunk_hist" in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"]); prev = t0
out = open("$R/gpurun_out/tl_$TAG.txt", "w")
for r in rows[s:e]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("uwie::(anonymous namespace)::", "").replace("uwie::", "").replace("void ", "")
    out.write(f"{(a - t0) / 1e3:8.1f} us  gap {(a - prev) / 1e3:6.1f}  dur {(b - a) / 1e3:7.1f}  {name[:70]}\n")
    prev = b
out.write(f"step: {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us, {e - s} kernels\n")
PY
tail -1 $R/gpurun_out/tl_$TAG.txt
