#!/bin/bash
# Counter passes over bench.py (2 steps) for the kernels of the step (GPU box).  usage: tools/pmc_bench.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/${TAG}_pmc1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" > $R/gpurun_out/${TAG}_pmc1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}_pmc2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" > $R/gpurun_out/${TAG}_pmc2.log 2>&1
find $R/gpurun_out/${TAG}_pmc1 $R/gpurun_out/${TAG}_pmc2 -type f ! -name '*counter_collection.csv' -delete
