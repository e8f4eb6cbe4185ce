#!/bin/bash
# A/B of library variants on the headline bench: bash tools/ab_lib.sh <outdir> <variant> [<variant> ...]   ("base" = the product library);
# two alternating repetitions of `bench.py --steps 40 --no-cpu-baseline` per variant.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset RECNOW_LIB_PATH; else export RECNOW_LIB_PATH=$R/rec_now_amd/librecnow_hip.$v.so; fi
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/${v}_$rep.json 2>> $O/err.log || exit 1
  done
done
unset RECNOW_LIB_PATH
python3 tools/benchsum.py $O/*.json
