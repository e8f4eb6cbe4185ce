"""Where the split-precision step's error comes from: scores and d loss / d x of the c3 model at 8192 rows (product route forced) in the two arithmetics
against the fp64 oracle -- max and rms error, and how many entries carry an error above half the maximum (a handful = a bug, a broad tail = rounding)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/tests', ROOT + '/oracle'):
    sys.path.insert(0, p)
os.environ['RECNOW_TILE'] = '0'
import pairs_oracle as PO
from _chunked_oracle import run_chunked, weights64
from test_northstar_gpu import _mix_fwd
from test_step_gpu import _model
from rec_now_amd import _lib
from rec_now_amd.step import DCNMixPairwiseStep
dev = torch.device('cuda:0')
B, D, S, N, L = 32768, 1024, 64, 2, 3
x, groups, labels, xd, yd, gd, cross, head = _model(dev, B, D, S, N, L, 4242)
named = dict(cross.named_weights()); named['head/kernel'], named['head/bias'] = head.kernel, head.bias
w64 = weights64(named)
fwd = _mix_fwd(w64, L, head=True)
(rs,), _, _ = run_chunked(fwd, torch.from_numpy(x), None, w64, chunk=4096, want_dx=False)
rloss, rds, rP = PO.pairwise_bpr(groups, labels, rs.astype(np.float32), grouped=True)
_, rdx, rgrads = run_chunked(fwd, torch.from_numpy(x), torch.from_numpy(rds), w64, chunk=4096)
step = DCNMixPairwiseStep(cross, head, xd, yd, gd)
for prec in (0, 1):
    _lib.call('recnow_set_gemm_precision', prec)
    step.run(); torch.cuda.synchronize()
    for name, got, ref in (('scores', step.scores, rs), ('dx', step.dx, rdx)):
        e = np.abs(got.detach().cpu().double().numpy() - ref)
        m = np.abs(ref).max()
        print('precision %d %-7s max err / max ref %.3g  rms err / rms ref %.3g  entries above half the max error: %d of %d; worst at %s'
              % (prec, name, e.max() / m, np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()), int((e > 0.5 * e.max()).sum()), e.size, np.unravel_index(e.argmax(), e.shape)))
_lib.call('recnow_set_gemm_precision', 0)
