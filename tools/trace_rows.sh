#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the step at a shard size.  usage: bash tools/trace_rows.sh <tag> <rows> [bench args]
TAG=$1; ROWS=$2; shift; shift
R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --rows $ROWS "$@" > $O/${TAG}_trace.log 2>&1
find $O/${TAG}_trace -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' -delete
for f in $(find $O/${TAG}_trace -name '*kernel_trace.csv'); do (head -n 1 $f; tail -n 900 $f) > $f.tail; rm $f; done
cd $R
