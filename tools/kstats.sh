#!/bin/bash
# Per-kernel statistics of a bench.py run (rocprofv3 --kernel-trace --stats): bash tools/kstats.sh <outdir under gpurun_out> <bench flags...> ; env switches come from the caller.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
find $O/prof -type f ! -name '*kernel_stats.csv' -delete
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print('%8.1f us x %4s = %6.2f %%  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], float(r['Percentage']), r['Name'][:150]))
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench.log | head -1
