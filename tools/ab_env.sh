#!/bin/bash
# A/B of environment switches on the headline bench: bash tools/ab_env.sh <outdir> "<bench flags>" NAME=VALUE [NAME=VALUE ...]   ("base" = no switch);
# two alternating repetitions of `bench.py --steps 40 --no-cpu-baseline <flags>` per setting.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
FLAGS=$1; shift
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then
      python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline $FLAGS > $O/${v}_$rep.json 2>> $O/err.log || exit 1
    else
      env "$v" python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline $FLAGS > $O/${v}_$rep.json 2>> $O/err.log || exit 1
    fi
  done
done
python3 tools/benchsum.py $O/*.json
