#!/bin/bash
# Fast iteration loop on the GPU box (through gpurun from the repo root):   bash profiles/quick.sh <tag> [bench args...]
#   1. bench line + per-kernel table of one recorded step (HIP events)  -> gpurun_out/<tag>_bench.log / _kernels.txt
#   2. one rocprofv3 PMC pass (SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES) over one step, folded per kernel into
#      wave-instructions and lane-instructions per frame pixel                    -> gpurun_out/<tag>_valu.txt
# QUICK_PMC=0 skips step 2.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-extras --kernel-table --steps 10 --warmup 2 "$@" > $R/gpurun_out/${TAG}_bench.log 2> $R/gpurun_out/${TAG}_kernels.txt || exit 1
python3 - <<PY
import json
for l in open("$R/gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("ms_per_step", d["ms_per_step"], "value", d["value"], "dominant", d["roofline"]["kernel"], d["roofline"]["kernel_ms_per_launch"], "frac", d["roofline"]["frac"])
PY
grep -E "^ *k_|ms" $R/gpurun_out/${TAG}_kernels.txt | head -60
[ "${QUICK_PMC:-1}" = "0" ] && exit 0
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 "$@" > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 $R/profiles/summarize_valu.py $OUT "$@" | tee $R/gpurun_out/${TAG}_valu.txt
