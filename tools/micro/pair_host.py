"""Host-side cost of one pairwise_loss forward+backward at BASELINE config 2 (B = 8192, 128 groups): enqueue time without synchronisation,
split into forward and backward, against the GPU time of the same step.  GPU only.  usage: python tools/micro/pair_host.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
B, G = 8192, 128
out = torch.randn(B, 1, device=dev, requires_grad=True)
lab = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)
grp = torch.from_numpy(rng.integers(0, G, B).astype(np.int32)).to(dev)
for _ in range(20):
    out.grad = None
    pairwise_loss(out, lab, grp).backward()
torch.cuda.synchronize()
N = 300
tf = tb = 0.0
t0 = time.perf_counter()
for _ in range(N):
    out.grad = None
    a = time.perf_counter()
    loss = pairwise_loss(out, lab, grp)
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    tf += b - a
    tb += c - b
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print('pairwise_loss c2: host forward %.1f us, host backward %.1f us, wall per step %.1f us' % (tf / N * 1e6, tb / N * 1e6, wall * 1e6))
if len(sys.argv) > 1 and sys.argv[1] == 'profile':
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        out.grad = None
        pairwise_loss(out, lab, grp).backward()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
