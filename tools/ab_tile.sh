#!/bin/bash
# GPU box: A/B of the row-block persistent forward (RECNOW_TILE) at the per-rank shard sizes.  bash tools/ab_tile.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ab_tile}
mkdir -p $O
cd $R
for rep in 1 2; do
  for rows in 8192 16384 32768; do
    for t in 0 1; do
      RECNOW_TILE=$t python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --rows $rows --force-dist > $O/r${rows}_tile${t}_$rep.json 2>> $O/err.log || exit 1
    done
  done
done
python3 tools/benchsum.py $O/*.json
