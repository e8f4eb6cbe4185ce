"""Turns the rocprofv3 outputs of tools/profile_configs.sh (gpurun_out/<tag>_cfg_*) into the committed per-config summaries:
profiles/<tag>_{fm_c4,dcn,cin_c4,ple_c5,pairwise_c2c3,listwise_c5}_kernel_stats.csv and profiles/<tag>_configs_summary.md, with the figure
north_star names for each row -- achieved HBM GB/s of the ALGORITHMIC bytes for the bandwidth-bound layers (FM, DCN; plus the PMC
bytes per launch), MFMA TFLOP/s of the algorithmic flops against the 157.3 TFLOP/s fp32 peak for the dense contractions (CIN, PLE),
rows/s for the ranking losses.   usage: python tools/make_config_profiles.py [tag]"""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
PEAK_HBM, PEAK_MFMA = 8000.0, 157.3
NAMES = {'fm': 'fm_c4', 'dcn': 'dcn', 'cin': 'cin_c4', 'ple': 'ple_c5', 'pair': 'pairwise_c2c3', 'list': 'listwise_c5',
         'embed': 'embed', 'senet': 'senet', 'ipnn': 'inner_pnn', 'attn': 'attention'}


def stats(key):
    f = sorted(glob.glob('gpurun_out/%s_cfg_%s/*/*_kernel_stats.csv' % (tag, key)), key=os.path.getmtime)[-1]
    shutil.copy(f, 'profiles/%s_%s_kernel_stats.csv' % (tag, NAMES[key]))
    return list(csv.DictReader(open(f)))


def pmc(key):
    out = {}
    for cnt in ('FETCH_SIZE', 'WRITE_SIZE'):
        fs = glob.glob('gpurun_out/%s_cfg_%s_%s/*/*_counter_collection.csv' % (tag, key, cnt))
        if not fs:
            return None
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(sorted(fs, key=os.path.getmtime)[-1])):
            if r['Counter_Name'] == cnt:
                agg[r['Kernel_Name']][0] += 1
                agg[r['Kernel_Name']][1] += float(r['Counter_Value'])
        out[cnt] = {k: v / n for k, (n, v) in agg.items()}
    # bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950: (2 * FETCH_SIZE + WRITE_SIZE) KB
    return {k: (2 * out['FETCH_SIZE'][k] + out['WRITE_SIZE'].get(k, 0.0)) * 1024 for k in out['FETCH_SIZE']}


def find(rows, prefix):
    for r in rows:
        if r['Name'].replace('void ', '').startswith(prefix):  # noqa: E501
            return r
    return None


lines = ['# Round %s -- rocprofv3 `--kernel-trace --stats` of every BASELINE config besides the c3 step (1x MI355X)\n' % tag[1:],
         'Produced by `tools/profile_configs.sh %s` (one rocprofv3 run per config of `tools/layer_bench.py 10 <config>`, eager launches; FETCH_SIZE / '
         'WRITE_SIZE in separate `--pmc` passes) and `tools/make_config_profiles.py`.  Raw per-kernel CSVs: `%s_<config>_kernel_stats.csv`.  '
         'Durations are rocprofv3 averages per launch; "algorithmic" bytes / flops are SURVEY.md section 8d\'s per-sample figures x the batch.\n' % (tag, tag)]

# ---- FM (c4, global batch) --------------------------------------------------------------------------------------------------
rows = stats('fm')
B, F, D = 131072, 64, 16
traffic = pmc('fm') or {}
lines.append('## FMLayer, configs[3] global batch: B = %d, F = %d, D = %d  (bound: HBM, %.0f GB/s spec)\n' % (B, F, D, PEAK_HBM))
lines.append('| kernel | avg us | algorithmic bytes / launch | achieved GB/s | of 8 TB/s | PMC bytes / launch (2 FETCH + WRITE) |\n|---|---|---|---|---|---|')
tot_us, tot_b = 0.0, 0.0
for pre, byts, what in (('k_fm_fwd', 4.0 * B * F * D + 8.0 * B, 'x read once, y and the saved field sum written'),
                        ('k_fm_bwd', 8.0 * B * F * D + 8.0 * B, 'x read, dx written, dy and the saved sum read')):
    r = find(rows, pre)
    us = float(r['AverageNs']) / 1e3
    tr = [v for k, v in traffic.items() if k.replace('void ', '').startswith(pre)]
    lines.append('| `%s` (%s) | %.1f | %.0f MB | %.0f | %.2f | %s |' % (r['Name'].split('(')[0].replace('void ', ''), what, us, byts / 1e6, byts / us / 1e3,
                                                                  byts / us / 1e3 / PEAK_HBM, ('%.0f MB' % (tr[0] / 1e6)) if tr else 'n/a'))
    tot_us += us
    tot_b += byts
lines.append('| forward + backward | %.1f | %.0f MB (12 B F D) | %.0f | %.2f | |\n' % (tot_us, tot_b / 1e6, tot_b / tot_us / 1e3, tot_b / tot_us / 1e3 / PEAK_HBM))

# ---- DCN-v1 -----------------------------------------------------------------------------------------------------------------
rows = stats('dcn')
B, D, L = 65536, 1024, 3
traffic = pmc('dcn') or {}
lines.append('## DCNLayer (v1), L = %d: B = %d, D = %d  (bound: HBM)\n' % (L, B, D))
lines.append('| kernel | avg us | algorithmic bytes / launch | achieved GB/s | of 8 TB/s | PMC bytes / launch |\n|---|---|---|---|---|---|')
tot_us, tot_b = 0.0, 0.0
for pre, byts, what in (('k_dcn_fwd', 8.0 * B * D, 'x read, y written (+ L scalars per row)'), ('k_dcn_bwd', 12.0 * B * D, 'x, dy read, dx written')):
    r = find(rows, pre)
    us = float(r['AverageNs']) / 1e3
    tr = [v for k, v in traffic.items() if k.replace('void ', '').startswith(pre)]
    lines.append('| `%s` (%s) | %.1f | %.0f MB | %.0f | %.2f | %s |' % (r['Name'].split('(')[0].replace('void ', ''), what, us, byts / 1e6, byts / us / 1e3,
                                                                  byts / us / 1e3 / PEAK_HBM, ('%.0f MB' % (tr[0] / 1e6)) if tr else 'n/a'))
    tot_us += us
    tot_b += byts
lines.append('| forward + backward | %.1f | %.0f MB (20 B D) | %.0f | %.2f | |\n' % (tot_us, tot_b / 1e6, tot_b / tot_us / 1e3, tot_b / tot_us / 1e3 / PEAK_HBM))


def gemm_table(rows, steps, alg_flops_step, title, note):
    lines.append(title)
    lines.append('| kernel | launches / step | avg us | ms / step | % of the step\'s kernel time |\n|---|---|---|---|---|')
    tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6
    gemm = 0.0
    for r in rows[:10]:
        ms = float(r['TotalDurationNs']) / steps / 1e6
        if 'k_gemm' in r['Name'] or 'k_cin_bwd_fused' in r['Name']:
            gemm += ms
        lines.append('| `%s` | %.1f | %.1f | %.3f | %.1f |' % (r['Name'].split('(')[0].replace('void ', '')[:80], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, ms, 100 * ms / tot))
    lines.append('\nKernel time per step %.2f ms, of which MFMA GEMM kernels %.2f ms.  Algorithmic flops per step (forward + dX + dW = 3 x forward) %.2f TFLOP: '
                 '**%.1f TFLOP/s of the algorithmic flops over the whole step = %.2f of the %.1f TFLOP/s fp32 MFMA peak** (%.1f TFLOP/s inside the GEMM kernels).  %s\n'
                 % (tot, gemm, alg_flops_step / 1e12, alg_flops_step / tot / 1e9, alg_flops_step / tot / 1e9 / PEAK_MFMA, PEAK_MFMA, alg_flops_step / gemm / 1e9, note))


# ---- CIN (c4 per rank) ------------------------------------------------------------------------------------------------------
rows = stats('cin')
B, F, D, Hs = 16384, 64, 16, [128, 128, 128]
ext = [F] + Hs
fwd = 2.0 * D * F * sum(ext[k - 1] * ext[k] for k in range(1, len(ext))) * B
steps = 13            # layer_bench: 10 warm-up + 3 timed steps (n = max(3, reps // 3)); the GEMM launch counts below divide evenly by it
GEMM_MS_CIN = sum(float(r['TotalDurationNs']) for r in rows if ('k_gemm' in r['Name'] or 'k_cin_bwd_fused' in r['Name']) and 'reduce' not in r['Name']) / steps / 1e6
gemm_table(rows, steps, 3 * fwd, '## CINLayer, configs[3] per-rank share: B = %d, F = %d, D = %d, H = %s  (bound: fp32 MFMA)\n' % (B, F, D, Hs),
           'Round 4 (VERDICT r3 item 4): the backward is TWO forward-sized products per layer, as SURVEY 8d prices it.  dW = dX_k^T Z stays a product of its own (K = B D rows, '
           'split-K); the data gradients share one product: T = dX_k W_k (`k_cin_bwd_fused`, csrc/cin_bwd.hip) is formed once on the matrix cores and BOTH reductions -- '
           'dX_{k-1} = sum_f T x0[f] into a second accumulator, dx0[f] = sum_h T X_{k-1}[h] by a 32-lane transpose-reduce -- run on the accumulator tile on the VALU.  '
           '(The round-3 text here said the three contractions could not be derived from one another; that was wrong: rounds 1-3 formed T twice.)  Executed = algorithmic = '
           '%.2f TFLOP per step, **%.1f TFLOP/s = %.2f of the fp32 MFMA peak inside the product kernels**.'
           % (3 * fwd / 1e12, 3 * fwd / GEMM_MS_CIN / 1e9, 3 * fwd / GEMM_MS_CIN / 1e9 / PEAK_MFMA))

# ---- PLE (c5 per rank) ------------------------------------------------------------------------------------------------------
rows = stats('ple')
B = 32768
# forward GEMM flops of PLELayer(3, [[512, 256], [256, 128]], 2, 1) on D_in = 4096 (tools/layer_bench.py counts 2 * B * numel of every 2-D / 3-D weight)
shared_l0 = 8 * (4096 * 512 + 512 * 256)            # 8 experts (3 tasks x 2 + 2 shared) of layer 0
l1 = 8 * (256 * 256 + 256 * 128)
gates = 4096 * (3 * 4 + 8) + 256 * (3 * 4)
fwd = 2.0 * B * (shared_l0 + l1 + gates)
gemm_table(rows, 13, 3 * fwd, '## PLELayer, configs[4] per-rank share: B = %d, D_in = 4096, 3 tasks  (bound: fp32 MFMA)\n' % B,
           'Flops from the layer\'s weight shapes (experts 4096->512->256 and 256->256->128, gates); approximate to a few per cent.')

# ---- pairwise / listwise ------------------------------------------------------------------------------------------------------
rows = stats('pair')
lines.append('## pairwise_loss, configs[1] (B = 8192, 128 groups) and configs[2] sizes (B = 65 536, 1024 groups), uniform and Zipf-skewed group sizes  (bound: latency / integer)\n')
lines.append('Four cases x 20 steps in one trace (layer_bench `pair`); per-kernel averages over all of them:\n')
lines.append('| kernel | launches | avg us |\n|---|---|---|')
for r in rows[:9]:
    lines.append('| `%s` | %s | %.1f |' % (r['Name'].split('(')[0].replace('void ', '')[:80], r['Calls'], float(r['AverageNs']) / 1e3))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e3
lines.append('\nKernel time of the 80 steps: %.0f us = %.0f us per loss forward + backward on average over the four cases (the c2 case: `k_group_pack_small` + '
             '`k_pair_one` + three small launches); per-case wall and graph-replay times: `%s_layer_bench.txt`.\n' % (tot, tot / 80, tag))
rows = stats('list')
lines.append('## listwise loss, configs[4] global size: B = 262 144, 4096 groups  (bound: latency)\n')
lines.append('| kernel | launches / step | avg us |\n|---|---|---|')
for r in rows[:8]:
    lines.append('| `%s` | %.0f | %.1f |' % (r['Name'].split('(')[0].replace('void ', '')[:80], int(r['Calls']) / 20, float(r['AverageNs']) / 1e3))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 20 / 1e3
lines.append('\nKernel time per step %.0f us = %.0f M rows/s of GPU time; the grouping (`k_group_mid`, one cooperative launch; round 4: the float ids 0..4095 are sorted by their integer images, 2 digit passes instead of 3) is %.0f %% of it.\n'
             % (tot, 262144 / tot, 100 * float(find(rows, 'k_group_mid')['TotalDurationNs']) / 20 / 1e3 / tot))
# ---- HBM-bound layers beside the hot path (round 5) --------------------------------------------------------------------
B, F, D, L = 131072, 64, 16, 50
P = F * (F - 1) // 2
try:
    lines.append('## SURVEY 8f rows (round 5): achieved HBM GB/s of the kernels\' own bytes (inputs read once, outputs written once), B = 131 072, F = 64, D = 16\n')
    lines.append('| kernel | avg us | bytes / launch | achieved GB/s | of 8 TB/s |\n|---|---|---|---|---|')
    for key, items in (('senet', (('k_senet_fused_fwd', 8.0 * B * F * D, 'x read, x2 written'), ('k_senet_fused_bwd', 12.0 * B * F * D, 'x, dout read, dx written'))),
                       ('ipnn', (('k_ipnn_fwd_gram', 4.0 * B * F * D + 4.0 * B * P, 'x read, P = 2016 pair products written'),
                                 ('k_ipnn_bwd_gram', 8.0 * B * F * D + 4.0 * B * P, 'dP, x read, dx written'))),
                       ('attn', (('k_attn_dot_v4<4, false>', 4.0 * B * L * D, 'user embeddings read (L = 50)'),
                                 ('k_attn_dot_v4<4, true>', 8.0 * B * L * D, 'user embeddings read, their gradient written')))):
        rows = stats(key)
        tus, tb = 0.0, 0.0
        for pre, byts, what in items:
            r = find(rows, pre)
            if r is None:
                continue
            us = float(r['AverageNs']) / 1e3
            tus += us
            tb += byts
            lines.append('| `%s` (%s) | %.1f | %.0f MB | %.0f | %.2f |' % (r['Name'].split('(')[0].replace('void ', '')[:60], what, us, byts / 1e6, byts / us / 1e3, byts / us / 1e3 / PEAK_HBM))
        if tus:
            lines.append('| %s forward + backward | %.1f | %.0f MB | %.0f | %.2f |' % (NAMES[key], tus, tb / 1e6, tb / tus / 1e3, tb / tus / 1e3 / PEAK_HBM))
    rows = stats('embed')
    lines.append('\n## Pooled embedding lookup (SURVEY 8f.2): B = 65 536 x C = 100 ids, T = 64, D = 16, V = 2^20, Zipf ids -- kernels of one forward + backward\n')
    lines.append('| kernel | launches / step | avg us | us / step |\n|---|---|---|---|')
    nstep = max(int(find(rows, 'k_embed_rows_chunks')['Calls']), 1)
    tot = 0.0
    for r in rows[:18]:
        if r['Name'].startswith('void at::') or 'rocclr' in r['Name']:
            continue
        per = float(r['TotalDurationNs']) / nstep / 1e3
        tot += per
        lines.append('| `%s` | %.1f | %.1f | %.1f |' % (r['Name'].split('(')[0].replace('void ', '')[:70], int(r['Calls']) / nstep, float(r['AverageNs']) / 1e3, per))
    lines.append('\n(`k_embed_pool_fwd_v4` and `k_slot_targets` are also launched by the forward-only timing loop of the harness: their per-step figures count both.)  '
                 'Listed kernels: %.0f us per step.\n' % tot)
except (IndexError, TypeError, KeyError) as e:      # a refresh without those traces
    lines.append('(no SURVEY 8f traces in this refresh: %s)\n' % e)
open('profiles/%s_configs_summary.md' % tag, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
