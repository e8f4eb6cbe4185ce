"""Does a step of the autograd route reach the device allocator?  Prints torch's device-allocation counters around ten steps of
`dcn_mix_score` + `pairwise_loss_fused` at the metric's shape (round 5: the step took 5.2 ms instead of 3.4 with a 34 ms hole in the rocprofv3 timeline
between the loss and the backward launches)."""
import os
import sys
import time

VAR = sys.argv[1] if len(sys.argv) > 1 else ''
if 'q' in VAR:
    os.environ['GPU_MAX_HW_QUEUES'] = '8'
if 'c' in VAR:
    os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda:0')
model = bench.Model()
x, groups, labels = bench.synth_batch(65536, 3)
xd, gd, yd = (torch.from_numpy(v).to(dev) for v in (x, groups, labels))
model(xd[:256])
xd.requires_grad_(True)
from rec_now_amd.fused import dcn_mix_score  # noqa: E402
from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss_fused  # noqa: E402
params = list(model.parameters())


from rec_now_amd.rec_block.pairwise_loss_from_batch import group_rows  # noqa: E402
side = torch.cuda.Stream(device=dev)
last = {}
if 'h' in VAR:
    from rec_now_amd import _lib
    lib = _lib.load()
    lib.recnow_prof_enable(64 * 40)
    lib.recnow_prof_sample_every(5)


def step():
    for p in params:
        p.grad = None
    xd.grad = None
    seg = None
    if 's' in VAR:
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            seg = group_rows(gd)
    scores = dcn_mix_score(model.cross, model.head, xd)
    if 's' in VAR:
        torch.cuda.current_stream().wait_stream(side)
    loss, n = pairwise_loss_fused(scores, yd, gd, reduce_mean=True, segments=seg)
    loss.backward()
    if 'l' in VAR:
        last['scores'], last['n'] = scores, n


for _ in range(3):
    step()
torch.cuda.synchronize()
s0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
t1 = time.perf_counter()
s1 = torch.cuda.memory_stats()
print('ms per step %.3f' % ((t1 - t0) * 100))
for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'reserved_bytes.all.current', 'reserved_bytes.all.peak', 'allocated_bytes.all.peak'):
    print(k, s0.get(k), '->', s1.get(k))
