#!/bin/bash
# A/B: 64-row tiles for the K = B weight-gradient products at every size (RECNOW_GEMM_BM64_KB=1) against only where 128-row tiles cannot fill the chip (default)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_bm64kb
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in 0 1; do
    RECNOW_GEMM_BM64_KB=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/rows65536_kb${v}_$rep.json 2>> $O/err.log || exit 1
    for rows in 16384 32768; do
      RECNOW_GEMM_BM64_KB=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --rows $rows --force-dist > $O/rows${rows}_kb${v}_$rep.json 2>> $O/err.log || exit 1
    done
  done
done
python3 tools/benchsum.py $O/rows*.json
