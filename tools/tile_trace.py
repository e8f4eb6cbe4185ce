"""GPU box: phase stamps of the row-block persistent kernels (csrc/dcnmix_tile.hip) from the diagnostic build
    python tools/build_variant.py tiletrace -DRN_TILE_TRACE
    RECNOW_LIB_PATH=rec_now_amd/librecnow_hip.tiletrace.so python tools/tile_trace.py [rows]
Prints, for workgroup 0 / wave 0, the microseconds each phase of each layer took (100 MHz wall clock).  The kernels run between
back-to-back bench-like steps so that the clocks are the ones of a running job."""
import ctypes
import os
import sys

os.environ['RECNOW_TILE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rec_now_amd import _lib  # noqa: E402
from rec_now_amd.fused import dcn_mix_score  # noqa: E402
from test_fused_gpu import _build  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
NODX = len(sys.argv) > 2 and sys.argv[2] == 'nodx'          # x as data: the backward chain without the O_l / dx streams
dev = torch.device('cuda:0')
x, xd, cross, head, w, hk, hb = _build(dev, B, 1024, 64, 2, 3, 1)
if NODX:
    xd = xd.detach()
gs = torch.from_numpy(np.random.default_rng(1).normal(size=B).astype(np.float32)).to(dev)
lib = _lib.load()
lib.recnow_debug_tile_trace.restype = ctypes.c_int
for rep in range(4):
    for _ in range(30):                  # warm clocks: back-to-back steps
        s = dcn_mix_score(cross, head, xd)
        s.backward(gs)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert lib.recnow_debug_tile_trace(buf) == 0
    t = [v / 100.0 for v in buf]
    out = []
    prev = t[0]
    for l in range(3):
        out.append('L%d: ' % l + ' '.join('%.1f' % (t[2 + 6 * l + i] - (t[2 + 6 * l + i - 1] if i else prev)) for i in range(5)))
        prev = t[2 + 6 * l + 4]
    print('fwd total %.1f us [GEMM1, partials+sync, B, C, D] | ' % (prev - t[0]) + ' | '.join(out))
    out = []
    prev = t[32]
    for l in (2, 1, 0):
        out.append('L%d: ' % l + ' '.join('%.1f' % (t[34 + 6 * l + i] - (t[34 + 6 * l + i - 1] if i else prev)) for i in range(5)))
        prev = t[34 + 6 * l + 4]
    print('bwd total %.1f us [GEMM2, partials+sync, c, d, g_l product] | ' % (prev - t[32]) + ' | '.join(out))
    print('fwd layer 1 phase C: mfma chain done +%.1f, stores issued +%.1f, before barrier +%.1f, after barrier +%.1f' % (
        t[20] - t[2 + 6 + 2], t[21] - t[20], t[22] - t[21], t[2 + 6 + 3] - t[22]))
