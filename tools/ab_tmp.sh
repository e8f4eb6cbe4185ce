R=$PWD; O=$R/gpurun_out
run() { tag=$1; shift; timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; grep -i "fault" $O/$tag.err; tail -3 $O/$tag.err; exit 1; }
python3 - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'par', d['parity_max_rel'], 'gemm us %.1f'%(r.get('avg_launch_us') or 0), {k:round(v['avg_launch_us'],1) for k,v in (r.get('hbm_bound_kernels') or {}).items()})
PY
}
for kt in 8 4; do
RECNOW_GEMM_MINKT=$kt RECNOW_MID_SLABS=0 run minkt${kt}_8192 --rows 8192 || exit 1
RECNOW_GEMM_MINKT=$kt RECNOW_MID_SLABS=0 run minkt${kt}_16384 --rows 16384 || exit 1
done
RECNOW_GEMM_MINKT=8 run slabs_8192 --rows 8192 || exit 1
