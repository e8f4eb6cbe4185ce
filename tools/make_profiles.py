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

tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)


def newest(pattern):
    fs = sorted(glob.glob(pattern) + glob.glob(pattern.replace('/*/', '/')), key=os.path.getmtime)
    return fs[-1]


import statistics


def trace_stats(trace_csv, out_csv=None):
    """Per-kernel statistics of the STEPS of a bench.py run from its rocprofv3 kernel trace: the dispatches before the first step's
    grouping launch (the lazy build of the layers on 256 rows, initialisers) are dropped, so an average is an average over launches of
    the benchmarked shape.  Returns rows sorted by total time: dicts Name, Calls, TotalDurationNs, AverageNs, MinNs, MaxNs, Percentage."""
    ev = []
    for r in csv.DictReader(open(trace_csv)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    ev.sort()
    # the first step starts with its grouping launch (side stream) next to k_pack_all (main stream): everything that started more than
    # 200 us before the first k_keys_* dispatch belongs to the lazy build
    # (round 5: the step's grouping launch forms the keys itself -- k_front_small / k_front_mid / k_group_mid<.., RAW> -- so any of the grouping kernels marks the spot;
    #  the lazy build on 256 rows runs no grouping)
    marks = ('k_keys_', 'void k_front_', 'void k_group_mid', 'void k_group_small')
    t_keys = next((e[0] for e in ev if e[2].startswith(marks)), ev[0][0])
    first = next((i for i, e in enumerate(ev) if e[0] >= t_keys - 200000), 0)
    agg = collections.OrderedDict()
    for s0, e0, n in ev[first:]:
        agg.setdefault(n, []).append(e0 - s0)
    tot = sum(sum(v) for v in agg.values())
    rows = [{'Name': n, 'Calls': len(v), 'TotalDurationNs': sum(v), 'AverageNs': sum(v) / len(v), 'MinNs': min(v), 'MaxNs': max(v),
             'Percentage': '%.2f' % (100.0 * sum(v) / tot)} for n, v in agg.items()]
    rows.sort(key=lambda r: -r['TotalDurationNs'])
    if out_csv:
        with open(out_csv, 'w') as fo:
            w = csv.writer(fo)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
            for r in rows:
                w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], '%.1f' % r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
    return rows, len(ev) - first, first


def bench_line(log):
    """The JSON line a profiled bench.py run printed (its own event hook's figures of the same run)."""
    for line in reversed(open(log, errors='replace').read().splitlines()):
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


f = newest('gpurun_out/%s_stats/*/*_kernel_trace.csv' % tag)
rows, n_disp, n_dropped = trace_stats(f, 'profiles/%s_bench_kernel_stats.csv' % tag)
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
        if k.startswith('void k_gemm_s3<'):       # round 6: the lean form of the split long-K product: one family with k_gemm_split (bench.py tag 8)
            other = ['k_gemm_split']
        if m or other:      # the persistent short-K kernel and the sub-space kernels are families of their own (bench.py *_TAGS)
            key = 'k_gemm<%s,%s,%s,%s>' % m.groups() if m else other[0]
            fam[key][0] += n
            fam[key][1] += n * b
traffic = {k: v / n for k, (n, v) in fam.items()}
# HBM bytes of one STEP: every dispatch of the counter run, (2 FETCH + WRITE) * 1024, over the steps it ran (one k_pack_all / k_tile_pack launch per step)
step_bytes = sum(n * (2 * fv + pmc['WRITE_SIZE'].get(k, (0, 0.0))[1]) * 1024 for k, (n, fv) in pmc['FETCH_SIZE'].items() if not k.startswith('void at::') and 'rocclr' not in k)
# (the lazy build of the layers on 256 rows takes the row-block route, i.e. k_tile_pack: the k_pack_all launches are the full-size steps)
n_steps_pmc = max(sum(n for k, (n, fv) in pmc['FETCH_SIZE'].items() if k.startswith('k_pack_all')), 1)
traffic_step_gb = step_bytes / n_steps_pmc / 1e9
sys.path.insert(0, ROOT)
from bench import kernel_source_hash      # noqa: E402  (bench.py refuses the table when the kernel sources have changed since)
json.dump({'kernel_source_sha256': kernel_source_hash(),
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 2; bytes = (2*FETCH_SIZE + '
                     'WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section (FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950); '
                     'launch-weighted mean over the instantiations of each tile family',
           'hbm_bytes_per_launch': traffic}, open('profiles/traffic.json', 'w'), indent=1)
hook = bench_line('gpurun_out/%s_stats.log' % tag)
clean = json.load(open('gpurun_out/%s_bench.json' % tag)) if os.path.exists('gpurun_out/%s_bench.json' % tag) else None
with open('profiles/%s_bench_summary.md' % tag, 'w') as fo:
    fo.write('# Round %s -- rocprofv3 --kernel-trace of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline` (1x MI355X)\n\n' % tag[1:])
    fo.write('Per-kernel CSV (computed from the kernel trace; the %d dispatches in front of the first step -- the lazy build of the layers on 256 rows '
             'and the initialisers -- are dropped, so every average is over launches of the benchmarked shape): `%s_bench_kernel_stats.csv`; HBM counters '
             '(separate --pmc passes): `%s_bench_pmc_hbm.csv`; bench line of the same build WITHOUT the profiler: `%s_bench.json`; GEMM '
             'microbenchmark (`tools/gemm_bench.py`): `%s_gemm_bench.txt`.\n\n' % (n_dropped, tag, tag, tag, tag))
    n_steps_trace = max(sum(r['Calls'] for r in rows if r['Name'].startswith('k_pack_all')), 1)
    fo.write('Sum of kernel durations: %.1f ms over %d steps (warm-up + timed + the untimed diagnostics: host-enqueue loop, exclusive-time account, parity step) = %.3f ms/step; '
             'the grouping of the batch runs on the main stream in front of the forward pass since round 5; `__amd_rocclr_copyBuffer` is the '
             'parity step copying results to the host, outside the timed region.\n\n' % (tot / 1e6, n_steps_trace, tot / 1e6 / n_steps_trace))
    fo.write('**HBM traffic of one step (PMC, `%s_bench_pmc_hbm.csv`): %.2f GB** = the sum over every dispatch of the counter run of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes, '
             'divided by its %d steps (algorithmic: 1.34 GB; round 4: 11.3 GB -- the input gradient is now written once by layer 0\'s product instead of as a '
             'read-modify-write in the product of every layer).\n\n' % (tag, traffic_step_gb, n_steps_pmc))
    spl = [r for r in rows if r['Name'].startswith(('void k_gemm_s3<', 'void k_gemm_split<'))]
    if spl:      # round 6: the headline step runs the six-term split products
        savg = sum(r['TotalDurationNs'] for r in spl) / sum(r['Calls'] for r in spl) / 1e3
        peak6 = 16 * 157.3 / 6
        fo.write('**The long-K products of the headline step (`k_gemm_s3<..>` / `k_gemm_split<..>`: fp32 operands as three bf16 pieces, six bf16 MFMA terms per product)**: '
                 'rocprofv3 average over their %d launches in this profiled run %.1f us = %.3f of %.1f TFLOP/s (dense bf16 peak / 6 terms; 17.45 GFLOP per launch)'
                 % (sum(r['Calls'] for r in spl), savg, 17.448 / savg / peak6 * 1e3, peak6))
        if hook and hook.get('roofline') and hook['roofline'].get('kernel') == 'k_gemm_split':
            fo.write('; the library\'s own HIP-event hook inside the same profiled run %.1f us = %.3f (step %.3f ms)' % (hook['roofline']['avg_launch_us'], hook['roofline']['frac'], hook['ms_per_step']))
        if clean and clean.get('roofline') and clean['roofline'].get('kernel') == 'k_gemm_split':
            fo.write('; the hook in the UNPROFILED run (`%s_bench.json`) %.1f us = %.3f, step %.3f ms' % (tag, clean['roofline']['avg_launch_us'], clean['roofline']['frac'], clean['ms_per_step']))
        fo.write('.  By instantiation: %s.\n\n' % '; '.join('`%s` %.1f us x %d' % (r['Name'][5:r['Name'].index('(')], float(r['AverageNs']) / 1e3, r['Calls']) for r in spl))
    gem = [r for r in rows if 'k_gemm<128, 128' in r['Name'] and ', 25>' not in r['Name']]      # (the ', 25>' instantiation: GEMM1 + sub-space forward, listed on its own)
    if gem:
        gavg = sum(r['TotalDurationNs'] for r in gem) / sum(r['Calls'] for r in gem) / 1e3
        fo.write('**`k_gemm<128,128,..>` (the dominant kernel), three figures of one build**: rocprofv3 average over its %d launches in this profiled run '
                 '%.1f us = %.3f of the 157.3 TFLOP/s fp32 MFMA peak (17.45 GFLOP per launch)' % (sum(r['Calls'] for r in gem), gavg, 17.448 / gavg / 157.3 * 1e3))
        if hook and hook.get('roofline'):
            fo.write('; the library\'s own HIP-event hook INSIDE the same profiled run %.1f us = %.3f (the hook agrees with rocprofv3: the profiler slows the run, '
                     'step %.3f ms here)' % (hook['roofline']['avg_launch_us'], hook['roofline']['frac'], hook['ms_per_step']))
        if clean and clean.get('roofline'):
            fo.write('; the hook in the UNPROFILED run of the same build (`%s_bench.json`, what `bench.py` reports) %.1f us = %.3f, step %.3f ms'
                     % (tag, clean['roofline']['avg_launch_us'], clean['roofline']['frac'], clean['ms_per_step']))
        fo.write('.\n\n')
        mf = [r for r in rows if 'k_gemm<128, 128' in r['Name'] and ', 25>' in r['Name']]
        if mf:
            fo.write('`k_gemm<128,128,..,25>` (the transposed GEMM1 with the sub-space forward in its epilogue, %d launches: its own kernel in these figures) %.1f us per '
                     'launch, of which the product itself is ~%.0f us; it replaces `k_gemm<..true, false..9>` + `k_mix_mid_fwd`.\n\n'
                     % (sum(r['Calls'] for r in mf), sum(r['TotalDurationNs'] for r in mf) / sum(r['Calls'] for r in mf) / 1e3, gavg))
    fo.write('| kernel | calls | total ms | avg us | min us | % |\n|---|---|---|---|---|---|\n')
    for r in rows[:26]:
        fo.write('| `%s` | %s | %.2f | %.1f | %.1f | %s |\n' % (r['Name'][:100].replace('|', '/'), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                        float(r['AverageNs']) / 1e3, r['MinNs'] / 1e3, r['Percentage']))
    # per-rank shard sizes of the 2/4/8-GPU rows
    fo.write('\n## The per-rank shards of the metric\'s 2 / 4 / 8-GPU rows on one GPU (`bench.py --rows R --force-dist`: every collective of the N > 1 path over a 1-rank RCCL group)\n\n')
    fo.write('| rows per GPU (N) | ms/step unprofiled | ideal = 1-GPU step / N | strong-scaling efficiency of the shard | host enqueue ms | kernel stats |\n|---|---|---|---|---|---|\n')
    base = None
    accounts = []
    for rws, n in ((65536, 1), (32768, 2), (16384, 4), (8192, 8), (8177, 8)):
        pth = 'gpurun_out/%s_bench_rows%d.json' % (tag, rws)
        if not os.path.exists(pth):
            continue
        d = json.loads(open(pth).read().strip().splitlines()[-1])
        shutil.copy(pth, 'profiles/%s_bench_rows%d.json' % (tag, rws))
        if base is None:
            base = d['ms_per_step']
        acct = (d.get('roofline') or {}).get('exclusive_ms_per_step')
        ks = ''
        try:
            ft = newest('gpurun_out/%s_stats_rows%d/*/*_kernel_trace.csv' % (tag, rws))
            trace_stats(ft, 'profiles/%s_bench_rows%d_kernel_stats.csv' % (tag, rws))
            ks = '`%s_bench_rows%d_kernel_stats.csv`' % (tag, rws)
        except (IndexError, StopIteration):
            pass
        fo.write('| %d (%d) | %.3f | %.3f | %.2f | %.3f | %s |\n' % (rws, n, d['ms_per_step'], base / n, base / n / d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], ks))
        if acct:
            accounts.append((rws, d.get('comm_exposed_ms'), acct))
    if accounts:
        fo.write('\nExclusive time per kernel family and step (ms; `roofline.exclusive_ms_per_step` of the lines above: ten untimed steps with every hooked launch '
                 'and phase recorded, an instant shared by k running launches counts 1/k for each) and what the collectives add to a step (`comm_exposed_ms`):\n\n')
        for rws, ce, acct in accounts:
            fo.write('* %d rows (comm exposed %s ms): %s\n' % (rws, ('%.3f' % ce) if ce is not None else '-', '; '.join('%s %.3f' % (k.split(' (')[0], v) for k, v in acct.items())))
    ph = 'gpurun_out/%s_bench_hash2.json' % tag
    if os.path.exists(ph):
        try:
            d = json.loads(open(ph).read().strip().splitlines()[-1])
            shutil.copy(ph, 'profiles/%s_bench_hash2.json' % tag)
            fo.write('\n`--gpus 2 --backend gloo --oversubscribe --shard hash --rows 32768` (ONE global batch of 65 536 rows split by `dp.shard_rows_by_group` over two '
                     'processes on this GPU): rows per rank %s, %.3f ms/step, cross-rank gate %s (worst %.2g).\n'
                     % (d['config']['rows_per_rank'], d['ms_per_step'], 'ok' if d['parity']['ok'] else 'FAILED', d['parity']['parity_max_rel']))
        except (ValueError, KeyError, IndexError):
            pass
    pg = 'gpurun_out/%s_bench_rows8192_graph.json' % tag
    if os.path.exists(pg):
        d = json.loads(open(pg).read().strip().splitlines()[-1])
        shutil.copy(pg, 'profiles/%s_bench_rows8192_graph.json' % tag)
        fo.write('\n8192 rows replayed from per-piece HIP graphs (`--graph`): %.3f ms/step, host %.3f ms.\n' % (d['ms_per_step'], d['config']['host_enqueue_ms_per_step']))
shutil.copy('gpurun_out/%s_bench.json' % tag, 'profiles/%s_bench.json' % tag)
try:      # the exact-fp32 step's own kernel trace
    trace_stats(newest('gpurun_out/%s_stats_f32/*/*_kernel_trace.csv' % tag), 'profiles/%s_bench_f32_kernel_stats.csv' % tag)
except (IndexError, StopIteration, ValueError):
    pass
for extra in ('bench_unfused', 'bench_autograd', 'bench_bf16x3', 'bench_f32', 'bench_c4', 'bench_c5', 'layer_bench'):
    for ext in ('json', 'txt'):
        src = 'gpurun_out/%s_%s.%s' % (tag, extra, ext)
        if os.path.exists(src):
            shutil.copy(src, 'profiles/%s_%s.%s' % (tag, extra, ext))
if os.path.exists('gpurun_out/%s_gemm_bench.txt' % tag):
    lines = [l for l in open('gpurun_out/%s_gemm_bench.txt' % tag) if 'TFLOP' in l]
    open('profiles/%s_gemm_bench.txt' % tag, 'w').writelines(lines)
print(json.dumps(traffic, indent=1))
print(open('profiles/%s_bench_summary.md' % tag).read()[:2500])
