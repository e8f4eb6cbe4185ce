"""As split_chain_numerics.py, on bench.py's own parity inputs (x * 120: saturated tanh, scores of O(0.3)): error of scores / dx per arithmetic against fp64,
and the same for the layer-by-layer activations x_1, x_2 of the forward pass (which layer / product the difference between the arithmetics enters at)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/tests', ROOT + '/oracle'):
    sys.path.insert(0, p)
import bench
import dense_ref as R
from rec_now_amd import _lib
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x, groups, labels = bench.synth_batch(B, 3, 0)
x = x * np.float32(bench.CHECK_SCALE)
torch.manual_seed(3)
model = bench.Model()
xd = torch.from_numpy(x).to(dev)
model(xd[:256])
named = {'cross.' + k: v for k, v in model.cross.named_weights().items()}
named['head.kernel'], named['head.bias'] = model.head.kernel, model.head.bias
w = {k: v.detach().cpu().double() for k, v in named.items()}
U, V, W, b, K, hk, hb = bench._split(w)
x64 = torch.from_numpy(x).double()
# fp64 layer by layer (dense_ref.dcn_mix_layer with one layer at a time)
refs = []
cur = x64
for l in range(3):
    outs = []
    for lo in range(0, B, 4096):
        outs.append(R.dcn_mix_layer_one(cur[lo:lo + 4096], x64[lo:lo + 4096], U[l], V[l], W[l], b[l], K[l]) if hasattr(R, 'dcn_mix_layer_one') else None)
    if outs[0] is None:
        break
    cur = torch.cat(outs)
    refs.append(cur.numpy())
from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
for prec in (0, 1):
    _lib.call('recnow_set_gemm_precision', prec)
    with torch.no_grad():
        y = model.cross(xd)
        s = model.head(y).reshape(-1)
    torch.cuda.synchronize()
    ry = np.concatenate([R.dcn_mix_layer(x64[lo:lo + 4096], U, V, W, b, K).numpy() for lo in range(0, B, 4096)])
    rs = np.concatenate([R.multi_dense_layer(torch.from_numpy(ry[lo:lo + 4096]), hk, hb).reshape(-1).numpy() for lo in range(0, B, 4096)])
    for name, got, ref in (('y (layer output)', y, ry), ('scores', s, rs)):
        e = np.abs(got.cpu().double().numpy() - ref)
        print('precision %d %-18s max err / max ref %.3g  rms err / rms ref %.3g  max|ref| %.3g  above half max: %d  worst at %s' % (prec, name, e.max() / np.abs(ref).max(),
              np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()), np.abs(ref).max(), int((e > 0.5 * e.max()).sum()), np.unravel_index(e.argmax(), e.shape)))
_lib.call('recnow_set_gemm_precision', 0)
# outliers of the split forward: where, and what the exact kernels give there
_lib.call('recnow_set_gemm_precision', 0)
with torch.no_grad():
    y0 = model.cross(xd).cpu().double().numpy()
_lib.call('recnow_set_gemm_precision', 1)
with torch.no_grad():
    y1 = model.cross(xd).cpu().double().numpy()
_lib.call('recnow_set_gemm_precision', 0)
e1 = np.abs(y1 - ry)
top = np.argsort(e1.reshape(-1))[-8:][::-1]
for t in top:
    r, c = np.unravel_index(t, e1.shape)
    print('row %5d col %4d: ref %+.7f split err %+.3g exact err %+.3g | row max |err| split %.3g exact %.3g | x0 there %+.4f' % (r, c, ry[r, c], y1[r, c] - ry[r, c], y0[r, c] - ry[r, c],
          e1[r].max(), np.abs(y0[r] - ry[r]).max(), x[r, c]))
rows = sorted(set(int(np.unravel_index(t, e1.shape)[0]) for t in top))
print('rows', rows, 'row % 128:', [r % 128 for r in rows])
# per-row error norm: are whole rows off (a row-level quantity: gate, T1 row) or single columns?
rowerr = e1.max(axis=1)
print('rows with max err > 2e-6 * max|ref|:', int((rowerr > 2e-6 * np.abs(ry).max()).sum()))
