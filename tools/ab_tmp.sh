R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
rm -rf $O/midf_stats$m
RECNOW_MIDF=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/midf_stats$m -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $O/midf_stats$m.log 2>&1
find $O/midf_stats$m -type f ! -name '*kernel_stats.csv' -delete
f=$(find $O/midf_stats$m -name '*kernel_stats.csv')
echo "== MIDF=$m"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r['Name']
    if 'k_gemm<' in n or 'mix_mid_fwd' in n:
        print('%-90s %4s %8.1f'%(n[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
