R=$PWD; O=$R/gpurun_out
run() { tag=$1; shift; timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; grep -i "fault" $O/$tag.err; tail -3 $O/$tag.err; exit 1; }
python3 - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'par', d['parity_max_rel'], {k:round(v['avg_launch_us'],1) for k,v in (r.get('hbm_bound_kernels') or {}).items()})
PY
}
for dbg in 0 1 2 3; do RECNOW_SK2=1 RECNOW_SK2_DBG=$dbg run sk2_dbg$dbg || exit 1; done
