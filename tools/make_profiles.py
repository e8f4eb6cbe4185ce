"""Turns the rocprofv3 outputs under gpurun_out/ (r01_stats, r01_pmc_fetch, r01_pmc_write, r01_bench.json) into the
committed summaries under profiles/.  usage: python tools/make_profiles.py [round_tag]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)


def newest(pattern):
    fs = sorted(glob.glob(pattern) + glob.glob(pattern.replace('/*/', '/')), key=os.path.getmtime)
    return fs[-1]


f = newest('gpurun_out/%s_stats/*/*_kernel_stats.csv' % tag)
shutil.copy(f, 'profiles/%s_bench_kernel_stats.csv' % tag)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
pmc = {}
for name, cnt in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = newest('gpurun_out/%s_pmc_%s/*/*_counter_collection.csv' % (tag, name))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != cnt:
            continue
        agg[r['Kernel_Name']][0] += 1
        agg[r['Kernel_Name']][1] += float(r['Counter_Value'])
    pmc[cnt] = {k: (n, v / n) for k, (n, v) in agg.items()}
fam = collections.defaultdict(lambda: [0, 0.0])
with open('profiles/%s_bench_pmc_hbm.csv' % tag, 'w') as fo:
    w = csv.writer(fo)
    w.writerow(['kernel', 'dispatches', 'FETCH_SIZE_avg_KB_raw', 'WRITE_SIZE_avg_KB', 'hbm_bytes_per_launch_corrected=(2*FETCH+WRITE)*1024'])
    for k, (n, fv) in sorted(pmc['FETCH_SIZE'].items(), key=lambda kv: -kv[1][1]):
        wv = pmc['WRITE_SIZE'].get(k, (0, 0.0))[1]
        b = (2 * fv + wv) * 1024
        w.writerow([k, n, '%.1f' % fv, '%.1f' % wv, '%.0f' % b])
        m = re.match(r'void k_gemm<(\d+), (\d+), (\d+), (\d+),', k)
        other = [f for f in ('k_gemm_shortk', 'k_mix_mid_fwd', 'k_mix_mid_bwd', 'k_gemm_split') if k.startswith('void %s<' % f) or k.startswith('void %s_fast<' % f)]
        if m or other:      # the persistent short-K kernel and the sub-space kernels are families of their own (bench.py *_TAGS)
            key = 'k_gemm<%s,%s,%s,%s>' % m.groups() if m else other[0]
            fam[key][0] += n
            fam[key][1] += n * b
traffic = {k: v / n for k, (n, v) in fam.items()}
sys.path.insert(0, ROOT)
from bench import kernel_source_hash      # noqa: E402  (bench.py refuses the table when the kernel sources have changed since)
json.dump({'kernel_source_sha256': kernel_source_hash(),
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 2; bytes = (2*FETCH_SIZE + '
                     'WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section (FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950); '
                     'launch-weighted mean over the instantiations of each tile family',
           'hbm_bytes_per_launch': traffic}, open('profiles/traffic.json', 'w'), indent=1)
steps = 19      # 3 warm-up + 10 timed + 5 host-enqueue diagnostic + 1 parity step
with open('profiles/%s_bench_summary.md' % tag, 'w') as fo:
    fo.write('# Round %s -- rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline` (1x MI355X)\n\n' % tag[1:])
    fo.write('Raw per-kernel CSV: `%s_bench_kernel_stats.csv`; HBM counters (separate --pmc passes): `%s_bench_pmc_hbm.csv`; bench line of the '
             'same build: `%s_bench.json`; GEMM microbenchmark (`tools/gemm_bench.py`): `%s_gemm_bench.txt`.\n\n' % (tag, tag, tag, tag))
    fo.write('Sum of kernel durations: %.1f ms over %d steps (3 warm-up + 10 timed + 5 untimed host-enqueue diagnostic + 1 parity step) = %.3f ms/step.  The batch-grouping kernels (sort, scans, '
             'segments) run on a side stream under the forward pass and are stretched by the GEMMs they share the chip with, so this sum is '
             'larger than the wall time per step (see `%s_bench.json`); `__amd_rocclr_copyBuffer` is the parity step copying d loss / d x and the '
             'gradients to the host, outside the timed region.\n\n' % (tot / 1e6, steps, tot / 1e6 / steps, tag))
    fo.write('| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|\n')
    for r in rows[:24]:
        fo.write('| `%s` | %s | %.2f | %.1f | %s |\n' % (r['Name'][:100].replace('|', '/'), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                   float(r['AverageNs']) / 1e3, r['Percentage']))
shutil.copy('gpurun_out/%s_bench.json' % tag, 'profiles/%s_bench.json' % tag)
for extra in ('bench_unfused', 'bench_forcedist', 'bench_bf16x3', 'layer_bench'):
    for ext in ('json', 'txt'):
        src = 'gpurun_out/%s_%s.%s' % (tag, extra, ext)
        if os.path.exists(src):
            shutil.copy(src, 'profiles/%s_%s.%s' % (tag, extra, ext))
try:
    shutil.copy(newest('gpurun_out/%s_stats_bf16x3/*/*_kernel_stats.csv' % tag), 'profiles/%s_bench_bf16x3_kernel_stats.csv' % tag)
except IndexError:
    pass
if os.path.exists('gpurun_out/%s_gemm_bench.txt' % tag):
    lines = [l for l in open('gpurun_out/%s_gemm_bench.txt' % tag) if 'TFLOP' in l]
    open('profiles/%s_gemm_bench.txt' % tag, 'w').writelines(lines)
print(json.dumps(traffic, indent=1))
print(open('profiles/%s_bench_summary.md' % tag).read()[:2500])
