"""The first synchronised step after an asynchronous warm-up of the autograd route: 36 ms in bench.py (round 5).  Allocator counters around it."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rec_now_amd.fused import dcn_mix_score  # noqa: E402
from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss_fused  # noqa: E402

dev = torch.device('cuda:0')
model = bench.Model()
x, groups, labels = bench.synth_batch(65536, 3)
xd, gd, yd = (torch.from_numpy(v).to(dev) for v in (x, groups, labels))
model(xd[:256])
xd.requires_grad_(True)
params = list(model.parameters())
last = {}


def step():
    for p in params:
        p.grad = None
    xd.grad = None
    scores = dcn_mix_score(model.cross, model.head, xd)
    loss, n = pairwise_loss_fused(scores, yd, gd, reduce_mean=True)
    loss.backward()
    last['scores'], last['n'] = scores, n
    return loss.detach()


keys = ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'reserved_bytes.all.current', 'active_bytes.all.current')
for _ in range(5):
    step()
print('after async warm-up', {k: torch.cuda.memory_stats().get(k) for k in keys})
torch.cuda.synchronize()
for i in range(6):
    c0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print('step %d: %.2f ms' % (i, (time.perf_counter() - c0) * 1e3), {k: torch.cuda.memory_stats().get(k) for k in keys})
