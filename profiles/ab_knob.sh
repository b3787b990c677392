#!/bin/bash
# A/B of one tuning knob's values inside ONE gpurun call (same library, environment variable UWIE_<KNOB> read by uwie_create):
#   gpurun -- 'bash profiles/ab_knob.sh ENTRY_FUSE "1 0 1 0" tag [bench args]'
# writes gpurun_out/<tag>_bench_<i>_<value>.log (the JSON line) and gpurun_out/<tag>_kern_<i>_<value>.txt (per-kernel table).
KNOB=$1; VALUES=$2; TAG=$3; shift 3
i=0
for v in $VALUES; do
  i=$((i+1))
  env UWIE_$KNOB=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --kernel-table --steps 10 --warmup 2 "$@" \
      > gpurun_out/${TAG}_bench_${i}_$v.log 2> gpurun_out/${TAG}_kern_${i}_$v.txt || exit 1
  echo "UWIE_$KNOB=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_${i}_$v.log)"
done
