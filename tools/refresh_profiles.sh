#!/bin/bash
# Run on the GPU box from the repository root (gpurun -- 'bash tools/refresh_profiles.sh r03'): writes the raw material of
# profiles/ into gpurun_out/; tools/make_profiles.py then turns it into the committed summaries.
# Every rocprofv3 line starts the program itself (python3), counters are collected in their own passes.
TAG=${1:-r06}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
PART=${2:-all}          # lines | prof | all (the whole refresh no longer fits one 20-minute gpurun call)
if [ "$PART" != prof ]; then
python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
python3 $R/bench.py --steps 20 --warmup 5 --unfused --no-cpu-baseline > $O/${TAG}_bench_unfused.json 2>> $O/${TAG}_bench.err
python3 $R/bench.py --steps 20 --warmup 5 --route autograd --no-cpu-baseline > $O/${TAG}_bench_autograd.json 2>> $O/${TAG}_bench.err
# round 6: the default line IS the split-precision step at this size (--gemm-precision auto); the exact-fp32 step as a line of its own (with the full fp64 gate),
# and the split line under its old name for the round-to-round comparison
python3 $R/bench.py --steps 20 --warmup 5 --gemm-precision f32 > $O/${TAG}_bench_f32.json 2>> $O/${TAG}_bench.err
cp $O/${TAG}_bench.json $O/${TAG}_bench_bf16x3.json
# the data-parallel model steps of configs[3] / configs[4] at their per-rank size, every collective over a 1-rank RCCL group
python3 $R/bench.py --config c4 --force-dist > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench.err
python3 $R/bench.py --config c5 --force-dist > $O/${TAG}_bench_c5.json 2>> $O/${TAG}_bench.err
# the per-rank shards of the metric's 1/2/4/8-GPU rows on one GPU, every collective of the N > 1 path over a 1-rank RCCL group
for rows in 65536 32768 16384 8192 8177; do       # (8177: a ragged per-rank batch on padded storage, recnow_dcn_mix_step_desc.B_pad)
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows $rows --force-dist > $O/${TAG}_bench_rows${rows}.json 2>> $O/${TAG}_bench.err || exit 1
done
# ONE global batch of 65 536 rows split by dp.shard_rows_by_group over two ranks (two processes on this one GPU over gloo): ragged shards, cross-rank gate
python3 $R/bench.py --gpus 2 --backend gloo --oversubscribe --shard hash --rows 32768 --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_hash2.json 2>> $O/${TAG}_bench.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows 8192 --force-dist --graph > $O/${TAG}_bench_rows8192_graph.json 2>> $O/${TAG}_bench.err
fi
if [ "$PART" = lines ]; then tail -c 300 $O/${TAG}_bench.json; exit 0; fi
cd /tmp && export TMPDIR=/tmp
for d in stats stats_f32 pmc_fetch pmc_write stats_rows32768 stats_rows16384 stats_rows8192; do rm -rf $O/${TAG}_$d; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other > $O/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_f32 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other --gemm-precision f32 > $O/${TAG}_stats_f32.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-other > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-other > $O/${TAG}_pmc_write.log 2>&1
for rows in 32768 16384 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_rows$rows -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other --rows $rows --force-dist > $O/${TAG}_stats_rows$rows.log 2>&1
done
# keep only the small CSVs (the merge back is capped at 64 MiB)
find $O/${TAG}_stats $O/${TAG}_stats_f32 $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_stats_rows* -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' -delete
find $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -name '*kernel_trace.csv' -delete
cd $R
python3 tools/gemm_bench.py > $O/${TAG}_gemm_bench.txt 2>&1
python3 tools/layer_bench.py 20 > $O/${TAG}_layer_bench.txt 2>&1
tail -c 400 $O/${TAG}_bench.json 2>/dev/null
