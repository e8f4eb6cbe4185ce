"""Diagnostic (GPU box): run the whole-step route EAGERLY with guard bands around every buffer the step owns and report any byte written
outside a buffer.  usage: python tools/guard_check.py ROWS [dist]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rec_now_amd import _lib, dp
from rec_now_amd.fused import GpuEvent
from rec_now_amd.step import DCNMixPairwiseStep

rows = int(sys.argv[1])
use_dist = len(sys.argv) > 2
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
if use_dist:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', device_id=dev)
    dp.FORCE_COLLECTIVES = True
GUARD = 1 << 20
guards = []
real_empty = torch.empty


def guarded(nbytes, device):
    n = max(int(nbytes), 16)
    n = (n + 255) // 256 * 256
    buf = torch.full((n + 2 * GUARD,), 0xA5, dtype=torch.uint8, device=device)
    guards.append((buf, n))
    return buf[GUARD:GUARD + n]


_lib.workspace = guarded
import rec_now_amd.step as S
S._lib.workspace = guarded
torch.manual_seed(3)
model = bench.Model()
x, groups, labels = bench.synth_batch(rows, 3, 0)
xd, gd, yd = (torch.from_numpy(v).to(dev) for v in (x, groups, labels))
model(xd[:256])
reducer = None
if use_dist:
    stages = DCNMixPairwiseStep.stages_for(model.cross, model.head)
    reducer = dp.LayerwiseReducer(stages, [GpuEvent() for _ in stages], dev)
    # re-home the buckets inside guard bands
    for i, flat in enumerate(reducer._flat):
        g = guarded(flat.numel() * 4, dev).view(torch.float32)[:flat.numel()]
        for p in reducer.stages[i]:
            v = reducer._view[id(p)]
            off = (v.data_ptr() - flat.data_ptr()) // 4
            reducer._view[id(p)] = g[off:off + v.numel()]
        reducer._flat[i] = g
st = DCNMixPairwiseStep(model.cross, model.head, xd, yd, gd, reducer=reducer)
print('ws', hex(st.ws.data_ptr()), st.ws.numel(), 'x', hex(st.x.data_ptr()), 'dx', hex(st.dx.data_ptr()), 'scores', hex(st.scores.data_ptr()), flush=True)
for _ in range(3):
    st.run()
torch.cuda.synchronize()
bad = 0
for buf, n in guards:
    lo, hi = buf[:GUARD], buf[GUARD + n:]
    for name, part in (('below', lo), ('above', hi)):
        idx = (part != 0xA5).nonzero()
        if idx.numel():
            bad += 1
            print('GUARD HIT %s buffer of %d bytes: %d bytes changed, first at offset %d, last %d' % (name, n, idx.numel(), int(idx[0]), int(idx[-1])), flush=True)
print('guard check rows=%d dist=%s: %s' % (rows, use_dist, 'CLEAN' if not bad else '%d hits' % bad))
