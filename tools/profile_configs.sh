#!/bin/bash
# GPU box: rocprofv3 kernel stats for every BASELINE config other than the c3 step (FM c4, DCN-v1, CIN c4, PLE c5, pairwise c2/c3,
# listwise c5) and the HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) for the two HBM-bound layers north_star names (FM, DCN).
# usage: bash tools/profile_configs.sh <tag>;   tools/make_config_profiles.py <tag> then writes profiles/<tag>_*.
TAG=${1:-r03}
R=$PWD; O=$R/gpurun_out
export RECNOW_LB_NOGRAPH=1
cd /tmp && export TMPDIR=/tmp
for key in fm dcn cin ple pair list embed senet ipnn attn; do
  rm -rf $O/${TAG}_cfg_$key
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cfg_$key -- python3 $R/tools/layer_bench.py 10 $key > $O/${TAG}_cfg_$key.log 2>&1 || exit 1
  find $O/${TAG}_cfg_$key -type f ! -name '*kernel_stats.csv' -delete
  grep -v "^W2\|^$" $O/${TAG}_cfg_$key.log | grep -i "fwd+bwd\|pairwise_loss\|listwise\|fwd " | head -6
done
for key in fm dcn; do
  for cnt in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${TAG}_cfg_${key}_$cnt
    rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/${TAG}_cfg_${key}_$cnt -- python3 $R/tools/layer_bench.py 3 $key > $O/${TAG}_cfg_${key}_$cnt.log 2>&1 || exit 1
    find $O/${TAG}_cfg_${key}_$cnt -type f ! -name '*counter_collection.csv' -delete
  done
done
cd $R
