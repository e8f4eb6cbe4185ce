#!/bin/bash
# GPU box: rocprofv3 kernel stats of the per-rank shard step (bench.py --rows R --force-dist), single stream (kernel times add up to the step) and
# with the second stream of the backward pass (the default).  usage: bash tools/trace_rows.sh <tag> <rows>
TAG=$1; ROWS=$2
R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ts in 0 1; do
  rm -rf $O/${TAG}_rows${ROWS}_ts$ts
  RECNOW_STEP_TWO_STREAMS=$ts rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_rows${ROWS}_ts$ts -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --rows $ROWS --force-dist > $O/${TAG}_rows${ROWS}_ts$ts.log 2>&1 || exit 1
  find $O/${TAG}_rows${ROWS}_ts$ts -type f ! -name '*kernel_stats.csv' -delete
done
cd $R
