"""Phase stamps of the sub-space backward kernel (k_mix_mid_bwd_fast) from a -DRN_MID_TRACE build:

    python tools/build_variant.py midtrace -DRN_MID_TRACE
    RECNOW_LIB_PATH=rec_now_amd/librecnow_hip.midtrace.so python tools/mid_trace.py

Runs two steps of the 65 536-row step and prints, for workgroup 0 and workgroup grid / 2, the microseconds between the stamps of every tile:
1 after the barrier that opens the tile | 2 gate math done | 3 next tile requested | 4 dA chain done (32 dependent MFMAs) | 5 dT1 stored |
6 dV chains done | 7 barrier | 8 gate columns stored | 9 next tile staged into LDS."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rec_now_amd import _lib  # noqa: E402
from rec_now_amd.step import DCNMixPairwiseStep  # noqa: E402

dev = torch.device('cuda:0')
model = bench.Model()
x, groups, labels = bench.synth_batch(65536, 3)
xd, gd, yd = (torch.from_numpy(v).to(dev) for v in (x, groups, labels))
model(xd[:256])
step = DCNMixPairwiseStep(model.cross, model.head, xd, yd, gd)
for _ in range(3):
    step.run()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * (2 * 16 * 10))()
rc = lib.recnow_debug_mid_trace(buf)
assert rc == 0, rc
for w in range(2):
    print('workgroup %s' % ('0' if w == 0 else 'grid/2'))
    base = buf[(w * 16) * 10 + 0]
    for k in range(6):
        t = [buf[(w * 16 + k) * 10 + i] for i in range(10)]
        if t[1] == 0:
            break
        print('  tile %d: opens at %7.2f us | ' % (k, (t[1] - base) / 100.0) + ' '.join('%d:%5.2f' % (i, (t[i] - t[i - 1]) / 100.0) for i in range(2, 10)))
