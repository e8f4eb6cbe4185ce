#!/bin/bash
# A/B of the paired A loads of k_gemm_s3 (RECNOW_S3_PAIR) on the headline bench + the PMC bytes of the long-K kernels.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/ab_env.sh abpair "--no-other" base RECNOW_S3_PAIR=0
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/abpair/pmc_${v}_$c
    RECNOW_S3_PAIR=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/abpair/pmc_${v}_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-other > /dev/null 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in (1, 0):
    tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for c, idx in (('FETCH_SIZE', 1), ('WRITE_SIZE', 2)):
        for f in glob.glob('gpurun_out/abpair/pmc_%d_%s/*/*counter_collection.csv' % (v, c)):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == c and 'k_gemm_s3' in r['Kernel_Name']:
                    k = r['Kernel_Name'][:40]
                    tot[k][idx] += float(r['Counter_Value'])
                    if idx == 1: tot[k][0] += 1
    for k, (n, fe, wr) in sorted(tot.items()):
        print('RECNOW_S3_PAIR=%d %-40s launches %3d  HBM bytes per launch (2 FETCH + WRITE) %.1f MB' % (v, k, n, (2 * fe + wr) * 1024 / max(n, 1) / 1e6))
PY
find gpurun_out/abpair -name '*.csv' -delete
