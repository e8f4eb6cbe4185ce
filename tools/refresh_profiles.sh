#!/bin/bash
# Run on the GPU box from the repository root (gpurun -- 'bash tools/refresh_profiles.sh r01'): writes the raw material of
# profiles/ into gpurun_out/; tools/make_profiles.py then turns it into the committed summaries.
# Every rocprofv3 line starts the program itself (python3), counters are collected in their own passes.
TAG=${1:-r02}
R=$PWD
mkdir -p $R/gpurun_out
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
python3 $R/bench.py --steps 20 --warmup 5 --unfused --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_unfused.json 2>> $R/gpurun_out/${TAG}_bench.err
python3 $R/bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_forcedist.json 2>> $R/gpurun_out/${TAG}_bench.err
python3 $R/bench.py --steps 20 --warmup 5 --gemm-precision bf16x3 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_bf16x3.json 2>> $R/gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_pmc_fetch $R/gpurun_out/${TAG}_pmc_write $R/gpurun_out/${TAG}_stats_bf16x3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_bf16x3 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-precision bf16x3 > $R/gpurun_out/${TAG}_stats_bf16x3.log 2>&1
# keep only the small CSVs (the merge back is capped at 64 MiB)
find $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_pmc_fetch $R/gpurun_out/${TAG}_pmc_write $R/gpurun_out/${TAG}_stats_bf16x3 -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete
cd $R
python3 tools/gemm_bench.py > gpurun_out/${TAG}_gemm_bench.txt 2>&1
python3 tools/layer_bench.py 10 > gpurun_out/${TAG}_layer_bench.txt 2>&1
tail -c 600 gpurun_out/${TAG}_bench.json
