"""GPU box: phase stamps of the split-precision row-block forward (csrc/dcnmix_tile_split.hip) from the diagnostic build
    python tools/build_variant.py tstrace -DRN_TILE_TRACE
    RECNOW_LIB_PATH=rec_now_amd/librecnow_hip.tstrace.so python tools/tile_split_trace.py [rows]
Prints, for workgroup 0 / wave 0 and the LAST block it ran, the microseconds each phase of each layer took (100 MHz wall clock); the kernel runs inside
back-to-back steps so that the clocks are the ones of a running job."""
import ctypes
import os
import sys

os.environ['RECNOW_TILE_SPLIT'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rec_now_amd import _lib  # noqa: E402
from rec_now_amd.step import DCNMixPairwiseStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device('cuda:0')
x, groups, labels = bench.synth_batch(B, 3, 0)
torch.manual_seed(3)
model = bench.Model()
xd = torch.from_numpy(x).to(dev)
model(xd[:256])
_lib.call('recnow_set_gemm_precision', 1)
step = DCNMixPairwiseStep(model.cross, model.head, xd, torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev))
assert step.route_code() == 2
lib = _lib.load()
lib.recnow_debug_tile_split_trace.restype = ctypes.c_int
for rep in range(3):
    for _ in range(20):
        step.run()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert lib.recnow_debug_tile_split_trace(buf) == 0
    t = [v / 100.0 for v in buf]
    out = []
    prev = t[0]
    for l in range(3):
        out.append('L%d: ' % l + ' '.join('%.1f' % (t[2 + 6 * l + i] - (t[2 + 6 * l + i - 1] if i else prev)) for i in range(5)))
        prev = t[2 + 6 * l + 4]
    print('block total %.1f us [GEMM1, partials+sync, B, C+split, D] | ' % (prev - t[0]) + ' | '.join(out))
_lib.call('recnow_set_gemm_precision', 0)
