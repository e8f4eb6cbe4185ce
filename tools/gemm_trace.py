"""Diagnostic: per-workgroup phase timestamps (start, operands staged, main loop done, epilogue done), shader clock and
XCC/CU/wave-slot placement of one recnow_gemm launch.  Needs a trace build of the library:
    RECNOW_TRACE=1 python rec_now_amd/csrc/build.py --force && python tools/gemm_trace.py <shape index of tools/gemm_bench.py>
(rebuild without RECNOW_TRACE afterwards; the default build has no trace code in the kernels)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '1', sys.argv[1]] if len(sys.argv) > 1 else [sys.argv[0], '1', '7']
dev = torch.device('cuda:0')
tr = torch.zeros(8192 * 8 + 64 * 4 * 64 * 2, dtype=torch.int64, device=dev)
import tools.gemm_bench as gb   # (module import only: its shape loop runs under __main__)
os.environ['RECNOW_GEMM_TRACE'] = str(tr.data_ptr())
gb.reps = 1
if int(sys.argv[2]) == len(gb.SHAPES):                 # the fused GEMM1 (transposed product + sub-space forward in its epilogue)
    s = ('midf', 128, gb.B)
    gb.run_midf()
else:
    s = gb.SHAPES[int(sys.argv[2])]
    gb.run(*s)
torch.cuda.synchronize()
raw = tr.cpu().numpy()
t = raw[:8192 * 8].reshape(-1, 8)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
st, pro, ml, end = [(t[:, i] - t0) / 100.0 for i in range(4)]     # us (100 MHz)
clk = (t[:, 6] - t[:, 5]) / np.maximum(t[:, 3] - t[:, 0], 1) * 100.0
print('shader clock MHz during blocks: mean %.0f  min %.0f max %.0f' % (clk.mean(), clk.min(), clk.max()))
hw = t[:, 4] & 0xffffffff
xcc = t[:, 4] >> 32
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)
print('blocks', len(t), 'span us', end.max())
print('prologue  us: mean %.2f p50 %.2f p90 %.2f' % ((pro - st).mean(), np.median(pro - st), np.percentile(pro - st, 90)))
print('mainloop  us: mean %.2f p50 %.2f p90 %.2f' % ((ml - pro).mean(), np.median(ml - pro), np.percentile(ml - pro, 90)))
print('epilogue  us: mean %.2f p50 %.2f p90 %.2f' % ((end - ml).mean(), np.median(end - ml), np.percentile(end - ml, 90)))
print('total/blk us: mean %.2f' % (end - st).mean())
key = xcc * 1000 + cu
u = np.unique(key)
print('distinct (xcc,cu):', len(u), ' blocks per cu mean', len(t) / len(u))
k0 = u[0]
sel = np.where(key == k0)[0]
order = sel[np.argsort(st[sel])]
print('timeline of one CU (start, prologue_done, mainloop_done, end) us; wave slot:')
for i in order[:24]:
    print('  blk %5d  %7.2f %7.2f %7.2f %7.2f  slot %d simd %d' % (i, st[i], pro[i], ml[i], end[i], hw[i] & 0xf, (hw[i] >> 4) & 3))

# second table (trace build): per-wave barrier arrival / departure per k-tile of the first 64 workgroups, in shader cycles
M, N = s[1], s[2]
nwg = len(t)
bar = raw[8 * nwg: 8 * nwg + 64 * 4 * 64 * 2].reshape(64, 4, 64, 2)
ok = bar[:, :, :, 1] != 0
if ok.any():
    nt = int(ok[0, 0].sum())
    arr, dep = bar[:, :, :nt, 0], bar[:, :, :nt, 1]
    wait = dep - arr                                   # cycles a wave sat in the barrier
    period = np.diff(dep, axis=2)                      # cycles between consecutive barrier departures = one k-tile
    print('k-tiles traced per workgroup:', nt)
    print('barrier wait per wave and k-tile, cycles: mean %.0f p50 %.0f p90 %.0f max %.0f' % (
        wait.mean(), np.median(wait), np.percentile(wait, 90), wait.max()))
    print('k-tile period, cycles: mean %.0f p50 %.0f p10 %.0f p90 %.0f' % (
        period.mean(), np.median(period), np.percentile(period, 10), np.percentile(period, 90)))
    skew = arr.max(axis=1) - arr.min(axis=1)           # first-to-last arrival inside a workgroup
    print('arrival skew inside a workgroup, cycles: mean %.0f p90 %.0f' % (skew.mean(), np.percentile(skew, 90)))
    w0 = 0
    print('workgroup 0, first 12 k-tiles: per wave (arrive - tile start, wait):')
    for k in range(1, min(nt, 13)):
        base = dep[w0, :, k - 1].min()
        print('  tile %2d ' % k + '  '.join('%6d/%5d' % (arr[w0, w, k] - base, wait[w0, w, k]) for w in range(4)))
