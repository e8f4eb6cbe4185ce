R=$PWD; O=$R/gpurun_out
run() { tag=$1; shift; timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; grep -i "fault" $O/$tag.err; tail -3 $O/$tag.err; exit 1; }
python3 - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'host %.3f'%d['config']['host_enqueue_ms_per_step'], 'graph', d['config']['hip_graph'], 'par', d['parity_max_rel'], 'gemm us %.1f'%(r.get('avg_launch_us') or 0))
PY
}
run plain_65536 || exit 1
for rows in 65536 32768 16384 8192; do run efd_$rows --rows $rows --force-dist || exit 1; done
run plain_8192 --rows 8192 || exit 1
