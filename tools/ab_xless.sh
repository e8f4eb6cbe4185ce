#!/bin/bash
# A/B of round 4's "x_{l+1} is not materialised" (RECNOW_XLESS=0 / 1) at the metric's batch and at the shard sizes; 64-row tile bound 128 / 256 at 32 768 rows.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_xless
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in 0 1; do
    RECNOW_XLESS=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/rows65536_x${v}_$rep.json 2>> $O/err.log || exit 1
    for rows in 8192 16384; do
      RECNOW_XLESS=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --rows $rows --force-dist > $O/rows${rows}_x${v}_$rep.json 2>> $O/err.log || exit 1
    done
  done
  for b in 128 256; do
    RECNOW_GEMM_BM64=$b python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --rows 32768 --force-dist > $O/rows32768_bm${b}_$rep.json 2>> $O/err.log || exit 1
  done
done
python3 tools/benchsum.py $O/rows*.json
