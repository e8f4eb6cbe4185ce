#!/bin/bash
# GPU box: the per-rank shard sizes of the metric's 1/2/4/8-GPU rows on ONE GPU (bench.py --rows R --force-dist), the host side
# of the 8192-row step, the HIP-graph diagnostics and a kernel trace of the 8192-row step.  usage: bash tools/rows_sweep.sh <tag>
TAG=${1:-r03}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
for rows in 65536 32768 16384 8192; do
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows $rows --force-dist > $O/${TAG}_rows${rows}_fd.json 2> $O/${TAG}_rows${rows}_fd.err || exit 1
  echo "rows $rows fd: $(python3 -c "import json;d=json.load(open('$O/${TAG}_rows${rows}_fd.json'));print(d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['parity_max_rel'])")"
done
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows 8192 --hostprof $O/${TAG}_hostprof_8192.txt > $O/${TAG}_rows8192_plain.json 2> $O/${TAG}_rows8192_plain.err || exit 1
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows 8192 --force-dist --hostprof $O/${TAG}_hostprof_8192_fd.txt > $O/${TAG}_rows8192_fd2.json 2>> $O/${TAG}_rows8192_fd.err || exit 1
for rows in 65536 8192; do
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows $rows --graph > $O/${TAG}_graph_rows${rows}.json 2> $O/${TAG}_graph_rows${rows}.err || exit 1
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows $rows --graph --group-inline > $O/${TAG}_graph_inline_rows${rows}.json 2> $O/${TAG}_graph_inline_rows${rows}.err || exit 1
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows $rows --group-inline > $O/${TAG}_inline_rows${rows}.json 2> $O/${TAG}_inline_rows${rows}.err || exit 1
  echo "rows $rows: graph $(python3 -c "import json;print(json.load(open('$O/${TAG}_graph_rows${rows}.json'))['ms_per_step'])") graph+inline $(python3 -c "import json;print(json.load(open('$O/${TAG}_graph_inline_rows${rows}.json'))['ms_per_step'])") eager+inline $(python3 -c "import json;print(json.load(open('$O/${TAG}_inline_rows${rows}.json'))['ms_per_step'])")"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_stats8192
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats8192 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --rows 8192 --force-dist > $O/${TAG}_stats8192.log 2>&1
find $O/${TAG}_stats8192 -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' -delete
# keep the trace of the last steps only (size)
for f in $(find $O/${TAG}_stats8192 -name '*kernel_trace.csv'); do tail -n 700 $f > $f.tail; head -n 1 $f > $f.head; rm $f; done
cd $R
