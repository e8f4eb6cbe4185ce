#!/bin/bash
# Counter passes over tools/gemm_bench.py for one shape index (GPU box).  usage: tools/pmc_gemm.sh <shape_index> <tag>
# Each --pmc set is its own run (kernel-trace only beside it), outputs under gpurun_out/<tag>_pmc{1,2,3}.
IDX=$1; TAG=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/${TAG}_pmc1 -- python3 $R/tools/gemm_bench.py 5 $IDX > $R/gpurun_out/${TAG}_pmc1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/${TAG}_pmc2 -- python3 $R/tools/gemm_bench.py 5 $IDX > $R/gpurun_out/${TAG}_pmc2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}_pmc3 -- python3 $R/tools/gemm_bench.py 5 $IDX > $R/gpurun_out/${TAG}_pmc3.log 2>&1
find $R/gpurun_out/${TAG}_pmc1 $R/gpurun_out/${TAG}_pmc2 $R/gpurun_out/${TAG}_pmc3 -type f ! -name '*counter_collection.csv' -delete
