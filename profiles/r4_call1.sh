#!/bin/bash
# round 4, GPU call: full GPU suite, then the bench's kernel table with the rank-counting sweep (default) and with the histogram sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest $R/tests -m gpu -x -q > $O/r4a_tests.log 2>&1; rc=$?
tail -4 $O/r4a_tests.log
[ $rc -ne 0 ] && { grep -E "^E|FAILED|Error" $O/r4a_tests.log | head -30; }
[ $rc -eq 124 ] && exit 1
for v in 1 0 1 0; do
  UWIE_RANK_SWEEP=$v timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-extras --kernel-table --steps 10 --warmup 2 > $O/r4a_bench_rank$v.log 2> $O/r4a_kernels_rank$v.txt || exit 1
  python3 - <<PY
import json
for l in open("$O/r4a_bench_rank$v.log"):
    if l.startswith("{"):
        d = json.loads(l); print("rank_sweep=$v ms_per_step", d["ms_per_step"], "value", d["value"])
PY
  grep -E "k_restore|k_lin_|k_rank|k_sel|k_guided_split|k_stretch_lab|k_clahe_apply" $O/r4a_kernels_rank$v.txt | head -14
done
