#!/bin/bash
# GPU box: A/B of LDS-DMA operand staging (RECNOW_GEMM_GLDS=1: k_gemm<..,false,false,..,0,0,41>) against register staging (k_gemm<..,9>) on the
# K = B weight-gradient products of the DCN-v2 step, then on the whole step.  usage: bash tools/ab_glds.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-glds}
mkdir -p $O
cd $R
for rep in 1 2; do
  python3 tools/gemm_bench.py 30 22 > $O/gemm_base_$rep.txt 2>&1 || exit 1
  RECNOW_GEMM_GLDS=1 python3 tools/gemm_bench.py 30 22 > $O/gemm_glds_$rep.txt 2>&1 || exit 1
done
tail -n 3 $O/gemm_*.txt
bash tools/ab_env.sh ${1:-glds}/step "" base RECNOW_GEMM_GLDS=1
