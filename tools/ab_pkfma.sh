#!/bin/bash
# A/B of the side product's packed FMAs (product build) against plain v_fma_f32 (variant built by
# `python tools/build_variant.py plainfma -DRN_SP_PLAIN_FMA`), at the current schedule.  GPU box.  Output: gpurun_out/r4_pkfma/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_pkfma
mkdir -p $O
V=$R/rec_now_amd/librecnow_hip.plainfma.so
cd $R
for rep in 1 2; do
  for idx in 12 18 20 22 23; do
    python3 tools/gemm_bench.py 50 $idx >> $O/pk_$rep.log 2>&1 || exit 1
    RECNOW_LIB_PATH=$V python3 tools/gemm_bench.py 50 $idx >> $O/plain_$rep.log 2>&1 || exit 1
  done
done
for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --steps 40 > $O/bench_pk_$rep.json 2>> $O/bench.err || exit 1
  RECNOW_LIB_PATH=$V python3 bench.py --no-cpu-baseline --steps 40 > $O/bench_plain_$rep.json 2>> $O/bench.err || exit 1
done
cd /tmp && export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM"
for idx in 18 20; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_pk_$idx -- python3 $R/tools/gemm_bench.py 5 $idx > $O/pmc_pk_$idx.log 2>&1 || exit 1
  RECNOW_LIB_PATH=$V rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_plain_$idx -- python3 $R/tools/gemm_bench.py 5 $idx > $O/pmc_plain_$idx.log 2>&1 || exit 1
done
find $O -type f -name '*.csv' ! -name '*counter_collection.csv' -delete
find $O -type f \( -name '*.db' -o -name '*.json' -a -path '*pmc*' \) -delete
echo ab done
