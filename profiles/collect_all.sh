#!/bin/bash
# One gpurun call: bench line, rocprofv3 kernel stats and the PMC traffic passes of the same command.
#   bash profiles/collect_all.sh <tag>
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python $R/bench.py > $R/gpurun_out/${TAG}_bench.log 2>&1
tail -1 $R/gpurun_out/${TAG}_bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_stats.log 2>&1
cd $R && bash profiles/collect_pmc.sh $TAG --steps 1 --warmup 1 > /dev/null 2>&1
find $R/gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -2
