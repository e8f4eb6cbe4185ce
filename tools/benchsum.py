"""Summarise bench.py JSON lines: python tools/benchsum.py gpurun_out/a.log gpurun_out/b.log ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        print('%-28s unreadable (%s)' % (f.split('/')[-1], e))
        continue
    r = d.get('roofline') or {}
    h = r.get('hbm_bound_kernels', {})
    print('%-28s %.3f ms/step  %.2f M/s | k_gemm %.1f us (%.3f) | shortk %.1f us %d GB/s | mid %.1f/%.1f us | parity %s'
          % (f.split('/')[-1], d['ms_per_step'], d['value'] / 1e6, r.get('avg_launch_us', 0), r.get('frac', 0),
             h.get('k_gemm_shortk', {}).get('avg_launch_us', 0), h.get('k_gemm_shortk', {}).get('achieved', 0),
             h.get('k_mix_mid_fwd', {}).get('avg_launch_us', 0), h.get('k_mix_mid_bwd', {}).get('avg_launch_us', 0), d.get('parity_max_rel')))
