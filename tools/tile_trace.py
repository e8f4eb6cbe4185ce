"""GPU box: phase stamps of the row-block persistent forward (csrc/dcnmix_tile.hip) from the diagnostic build
    python tools/build_variant.py tiletrace -DRN_TILE_TRACE
    RECNOW_LIB_PATH=rec_now_amd/librecnow_hip.tiletrace.so python tools/tile_trace.py [rows]
Prints, for workgroup 0 / wave 0, the microseconds each phase of each layer took (100 MHz wall clock)."""
import ctypes
import os
import sys

os.environ['RECNOW_TILE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import torch  # noqa: E402

from rec_now_amd import _lib  # noqa: E402
from rec_now_amd.fused import dcn_mix_score  # noqa: E402
from test_fused_gpu import _build  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device('cuda:0')
x, xd, cross, head, w, hk, hb = _build(dev, B, 1024, 64, 2, 3, 1)
lib = _lib.load()
for rep in range(4):
    with torch.no_grad():
        s = dcn_mix_score(cross, head, xd.detach())
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.recnow_debug_tile_trace.restype = ctypes.c_int
    assert lib.recnow_debug_tile_trace(buf) == 0
    t = [v / 100.0 for v in buf]
    names = ['GEMM1 loop', 'partials+sync', 'phase B', 'phase C', 'phase D']
    line = []
    prev = t[0]
    for l in range(3):
        for i, n in enumerate(names):
            cur = t[2 + 6 * l + i]
            line.append('%s %.1f' % (n if l == 0 else n.split()[0][:5], cur - prev))
            prev = cur
    print('rep %d: total %.1f us | ' % (rep, prev - t[0]) + ' | '.join(line))
