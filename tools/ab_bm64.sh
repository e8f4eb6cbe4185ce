#!/bin/bash
# Small-M dispatch A/B: the per-rank shards of the metric's 4- and 8-GPU rows with the long-K products on 128-row tiles
# (RECNOW_GEMM_BM64=0), on 64-row tiles up to 64 / 128 tiles of 128 rows (=64: the 8192-row shard only; =128: the 16 384-row shard too).  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_bm64
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in 0 64 128; do
    for rows in 8192 16384; do
      RECNOW_GEMM_BM64=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --rows $rows --force-dist > $O/rows${rows}_bm${v}_$rep.json 2>> $O/err.log || exit 1
    done
  done
done
python3 tools/benchsum.py $O/rows*.json
