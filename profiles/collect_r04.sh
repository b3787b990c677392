#!/bin/bash
# Evidence of one round at its final HEAD, in ONE gpurun call (through gpurun from the repo root):
#     bash profiles/collect_r04.sh <tag>          e.g. r04_final
#   gpurun_out/<tag>_4k64_bench.log          the default bench line (headline workload, cpu_baseline, extras)
#   gpurun_out/<tag>_4k64_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_4k64_pmc_summary.txt    per-kernel HBM bytes (FETCH_SIZE x2 + WRITE_SIZE, separate passes) and SQ counters
#   gpurun_out/<tag>_traffic.json            the table bench.py's roofline.traffic reads (copy to profiles/r04_traffic.json)
#   gpurun_out/<tag>_uniform_*               the same kernel stats / SQ summary for --dist uniform (dense Canny maps)
#   gpurun_out/<tag>_1080p_batch1_timeline.txt, <tag>_secondary_4k16.txt, <tag>_guided_sq_summary.txt, <tag>_exact_mode_4k64.txt, <tag>_stream_4k.txt
# The raw counter CSVs (tens of MB per pass) are folded on the box and deleted: gpurun copies back at most 64 MiB.
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout -k 10 600 python $R/bench.py > $O/${TAG}_4k64_bench.log 2>&1
tail -1 $O/${TAG}_4k64_bench.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for dist in underwater uniform; do
  D=/tmp/stats_$dist; rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $R/bench.py --no-cpu-baseline --no-extras --dist $dist > $O/${TAG}_${dist}_stats.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  name=4k64; [ $dist = uniform ] && name=uniform_4k64
  [ -n "$f" ] && cp $f $O/${TAG}_${name}_kernel_stats.csv
  rm -rf $D
done
# PMC passes (headline workload): 3 steps per pass (warm-up, recorded, timed)
P=/tmp/pmc_$TAG; rm -rf $P; mkdir -p $P
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > $P.$name.log 2>&1; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 $R/profiles/summarize_pmc.py $P $((3 * 64 * 2160 * 3840)) $O/${TAG}_traffic.json 3 > $O/${TAG}_4k64_pmc_summary.txt 2>&1
python3 $R/profiles/summarize_valu.py $P/sq > $O/${TAG}_4k64_valu_per_px.txt 2>&1
rm -rf $P
# uniform noise: VALU / LDS instruction counts per kernel
P=/tmp/pmcu_$TAG; rm -rf $P
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $P -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --dist uniform > $P.log 2>&1
python3 $R/profiles/summarize_valu.py $P > $O/${TAG}_uniform_4k64_valu_per_px.txt 2>&1
rm -rf $P
cd $R
bash profiles/timeline_1080p.sh ${TAG} > /dev/null 2>&1
cp $O/tl_${TAG}.txt $O/${TAG}_1080p_batch1_timeline.txt; rm -rf $O/tl_${TAG}
python3 profiles/time_strategies.py > $O/${TAG}_secondary_4k16.txt 2>&1
# round 4: the SQ issue / stall / LDS counters of the guided kernel (the SQ_ACTIVE_INST_* summary belongs in profiles/),
# the exact-order filter's per-kernel times, the streaming mode's line (configs[4], one GPU) in both number formats
KFILTER=guided bash profiles/pmc_sq.sh ${TAG}_guided --no-extras > /dev/null 2>&1
python3 profiles/summarize_sq.py $O/sq_${TAG}_guided guided > $O/${TAG}_guided_sq_summary.txt 2>&1
rm -rf $O/sq_${TAG}_guided $O/sq_${TAG}_guided.*.log
python3 profiles/time_exact.py 2160 3840 64 > $O/${TAG}_exact_mode_4k64.txt 2>&1
for inter in f64 f32t; do
  timeout -k 10 300 python bench.py --mode stream --batch 96 --chunk 8 --steps 3 --warmup 1 --inter $inter 2>/dev/null | tail -1 >> $O/${TAG}_stream_4k.txt
done
ls -la $O | grep ${TAG}
