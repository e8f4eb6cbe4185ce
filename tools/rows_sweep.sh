#!/bin/bash
# GPU box: the per-rank shard sizes of the metric's 1/2/4/8-GPU rows on ONE GPU (bench.py --rows R --force-dist: every collective of
# the N > 1 path runs over a 1-rank RCCL group).  usage: bash tools/rows_sweep.sh <tag> [extra bench args]
TAG=${1:-r03}; shift
R=$PWD
O=$R/gpurun_out
mkdir -p $O
line() { python3 - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'host %.3f'%d['config']['host_enqueue_ms_per_step'], 'graph', d['config']['hip_graph'], 'par', d['parity_max_rel'],
      'gemm us %.1f'%(r.get('avg_launch_us') or 0), {k:round(v['avg_launch_us'],1) for k,v in (r.get('hbm_bound_kernels') or {}).items()})
PY
}
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/${TAG}_rows65536_plain.json 2> $O/${TAG}_rows65536_plain.err || exit 1
line $O/${TAG}_rows65536_plain.json
for rows in 65536 32768 16384 8192; do
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows $rows --force-dist "$@" > $O/${TAG}_rows${rows}_fd.json 2> $O/${TAG}_rows${rows}_fd.err || exit 1
  line $O/${TAG}_rows${rows}_fd.json
done
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows 8192 --force-dist --eager "$@" > $O/${TAG}_rows8192_fd_eager.json 2> $O/${TAG}_rows8192_fd_eager.err || exit 1
line $O/${TAG}_rows8192_fd_eager.json
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows 8192 --graph "$@" > $O/${TAG}_rows8192_graph.json 2> $O/${TAG}_rows8192_graph.err || exit 1
line $O/${TAG}_rows8192_graph.json
