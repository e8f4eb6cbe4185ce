"""Where the c3 step's parity budget goes: bench.py's own parity step (B rows, x * 120, seed 3) against the fp64 oracle of the whole batch, for both
arithmetics of the products in ONE process (the oracle is evaluated once and cached under /tmp).  Prints, per arithmetic and tensor: max and rms error
relative to max|ref|, how many entries carry more than half the maximum error (a handful = outliers, thousands = a tail), the worst locations, and the
error of the worst ROWS split into what the row's own scale would allow.  GPU only; diagnostic, not part of any gate.

    python tools/dx_error_probe.py [rows]          (RECNOW_SPLIT_SHORTK=0 / RECNOW_SPLIT_LONGK=0 in the environment select the families)
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rec_now_amd import _lib
from rec_now_amd.step import DCNMixPairwiseStep

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.set_num_threads(2 * bench.cpu_share())
x, groups, labels = bench.synth_batch(B, 3, 0)
torch.manual_seed(3)
model = bench.Model()
xd = torch.from_numpy(x).to(dev)
yd, gd = torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev)
model(xd[:256])
named = {'cross.' + k: v for k, v in model.cross.named_weights().items()}
named['head.kernel'], named['head.bias'] = model.head.kernel, model.head.bias
named_np = {k: v.detach().cpu().numpy() for k, v in named.items()}
xq_np = x * np.float32(bench.CHECK_SCALE)
cache = '/tmp/dx_probe_oracle_%d.npz' % B
if os.path.exists(cache):
    z = np.load(cache)
    r_dx, r_sc = z['dx'], z['scores']
    r_grads = {k: z['g_' + k.replace('/', '|')] for k in named_np}
else:
    t0 = time.time()
    _, r_dx, r_grads, r_sc, _, _ = bench.cpu_step_full(xq_np, groups, labels, named_np, dtype=torch.float64)
    print('fp64 oracle: %.1f s' % (time.time() - t0), flush=True)
    np.savez(cache, dx=r_dx, scores=r_sc, **{'g_' + k.replace('/', '|'): v for k, v in r_grads.items()})
pstep = DCNMixPairwiseStep(model.cross, model.head, xd.detach(), yd, gd, need_dx=True)
pstep.x.copy_(xd * bench.CHECK_SCALE)
m_dx = np.abs(r_dx).max()
rowmax = np.abs(r_dx).max(axis=1)
for prec in (0, 1):
    _lib.call('recnow_set_gemm_precision', prec)
    for p in named.values():
        p.grad = None
    pstep.run()
    torch.cuda.synchronize()
    dx = pstep.dx.cpu().numpy().astype(np.float64)
    sc = pstep.scores.cpu().numpy().astype(np.float64)
    e = np.abs(dx - r_dx)
    print('== precision %d (%s)  SPLIT_SHORTK=%s SPLIT_LONGK=%s' % (prec, 'bf16x3' if prec else 'f32', os.environ.get('RECNOW_SPLIT_SHORTK', '1'), os.environ.get('RECNOW_SPLIT_LONGK', '1')))
    print('scores: max err / max ref %.3g' % (np.abs(sc - r_sc).max() / np.abs(r_sc).max()))
    print('dx: max err / max ref %.4g   rms err / rms ref %.3g   max|ref| %.3g   entries above half the max error: %d, above a quarter: %d of %d'
          % (e.max() / m_dx, np.sqrt((e ** 2).mean()) / np.sqrt((r_dx ** 2).mean()), m_dx, int((e > 0.5 * e.max()).sum()), int((e > 0.25 * e.max()).sum()), e.size))
    flat = np.argsort(e.reshape(-1))[-6:][::-1]
    for t in flat:
        r, c = np.unravel_index(t, e.shape)
        print('   row %5d col %4d: ref %+.6e err %+.3e = %.3g of max|ref|, %.3g of the row max %.3e; row rms err %.3g of max|ref|; score ref %+.4f err %+.2e'
              % (r, c, r_dx[r, c], dx[r, c] - r_dx[r, c], e[r, c] / m_dx, e[r, c] / rowmax[r], rowmax[r], np.sqrt((e[r] ** 2).mean()) / m_dx, r_sc[r], sc[r] - r_sc[r]))
    # error relative to the ROW's own largest entry: is the worst row just the largest row?
    rr = e.max(axis=1) / np.maximum(rowmax, 1e-300)
    big = rowmax > 0.1 * m_dx
    print('   rows with max|dx| > 0.1 max: %d; their worst row-relative error %.3g; all rows (row max > 1e-3 max): worst row-relative %.3g'
          % (int(big.sum()), rr[big].max() if big.any() else 0.0, rr[rowmax > 1e-3 * m_dx].max()))
    for k in sorted(named):
        g = named[k].grad.cpu().numpy().astype(np.float64)
        ref = r_grads[k]
        s = np.abs(r_grads['head.kernel']).max() if k == 'head.bias' else np.abs(ref).max()
        print('   %-44s %.3g' % (k, np.abs(g - ref).max() / s))
_lib.call('recnow_set_gemm_precision', 0)
