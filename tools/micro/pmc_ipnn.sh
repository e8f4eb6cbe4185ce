#!/bin/bash
# per-kernel PMC passes for the InnerPNN kernels (one counter group per run)
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "FETCH_SIZE WRITE_SIZE" "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm_$tag -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py 2 ipnn > /dev/null 2>&1
  python3 - "$grp" /tmp/pm_$tag <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[2] + '/*/*counter_collection.csv')
if not f: print('no output for', sys.argv[1]); sys.exit()
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if 'ipnn' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'][:28], r['Counter_Name']); agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()): print(k[0], k[1], '%.4g' % (v / n))
PY
done
