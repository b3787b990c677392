#!/bin/bash
# rocprofv3 kernel averages of one strategy at 4K x 16: bash profiles/trace_strategy.sh <strategy 1..6>
set -e
K=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$K -- python3 $R/bench.py --strategy $K --batch 16 --no-extras --no-cpu-baseline --steps 5 > $R/gpurun_out/st_$K.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/st_$K/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    n = r["Name"]
    if "at::" in n: continue
    print(f"{float(r['TotalDurationNs'])/1e6/7:8.3f} ms/step  avg {float(r['AverageNs'])/1e3:9.1f} us  calls/step {int(r['Calls'])/7:5.1f}  " + n.replace("uwie::(anonymous namespace)::","").replace("uwie::","").replace("void ","")[:60])
PY
