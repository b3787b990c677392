#!/bin/bash
# round 4 iteration loop on the GPU box:  bash profiles/r4_quick.sh <tag> "<pytest -k expression or empty>" [ENV=VAL ...] : selected GPU tests, then the
# bench's per-kernel table once per listed environment setting ("-" = none)
TAG=$1; KEXPR=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout -k 10 600 python -m pytest $R/tests -m gpu -x -q -k "$KEXPR" > $O/${TAG}_tests.log 2>&1; rc=$?
  tail -3 $O/${TAG}_tests.log
  if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" $O/${TAG}_tests.log | head -30; exit 1; fi
fi
[ $# -eq 0 ] && set -- -
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-extras --kernel-table --steps 10 --warmup 2 > $O/${TAG}_bench$i.log 2> $O/${TAG}_kernels$i.txt || { tail -5 $O/${TAG}_kernels$i.txt; exit 1; }
  python3 - <<PY
import json
for l in open("$O/${TAG}_bench$i.log"):
    if l.startswith("{"):
        d = json.loads(l); print("[$setting] ms_per_step", d["ms_per_step"], "value", d["value"])
PY
  grep -E "^ +[0-9.]+ ms" $O/${TAG}_kernels$i.txt | head -${QUICK_LINES:-12}
done
