"""GPU box: phase stamps of the one-workgroup grouping kernel (csrc/scan_sort.hip k_group_small) from the diagnostic build
    python tools/build_variant.py gstrace -DRN_GS_TRACE
    RECNOW_LIB_PATH=rec_now_amd/librecnow_hip.gstrace.so python tools/gs_trace.py [rows] [groups]
Prints the microseconds between the stamps of thread 0 (100 MHz wall clock): load | varying bits | sort | heads + scan | writes."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rec_now_amd import _lib  # noqa: E402
from rec_now_amd.rec_block.pairwise_loss_from_batch import group_rows  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0')
ids = torch.from_numpy(np.random.default_rng(0).integers(0, G, B).astype(np.float32)).to(dev)
lib = _lib.load()
lib.recnow_debug_gs_trace.restype = ctypes.c_int
for rep in range(5):
    for _ in range(20):
        seg = group_rows(ids)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    assert lib.recnow_debug_gs_trace(buf) == 0
    t = [v / 100.0 for v in buf[:6]]
    if B > 8192:
        gm = (ctypes.c_longlong * 64)()
        lib.recnow_debug_gm_trace.restype = ctypes.c_int
        n = lib.recnow_debug_gm_trace(gm)
        t = [v / 100.0 for v in gm[:n]]
        print('rows %d groups %d k_group_mid workgroup 0: total %.1f us; phases (work | barrier alternating from phase 0): ' % (B, G, t[-1] - t[0]) +
              ' '.join('%.1f' % (t[i + 1] - t[i]) for i in range(n - 1)))
        continue
    print('rows %d groups %d varying %#x: total %.1f us | load %.1f | varying bits %.1f | sort %.1f | heads + scan %.1f | writes %.1f'
          % (B, G, buf[6], t[5] - t[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]))
