"""Probe: does the c3 step run faster as two half-batch steps on two streams (MFMA-bound products of one half beside the HBM-bound
products of the other) than as one full-batch step?  Timing only (each half computes its own in-batch loss)."""
import os
import sys
import time

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
from rec_now_amd.step import DCNMixPairwiseStep

dev = torch.device('cuda:0')
D, S, N, L = 1024, 64, 2, 3


def make(B, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.normal(0, 0.7, (B, D)).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.integers(0, B // 64, B).astype(np.float32)).to(dev)
    y = torch.from_numpy((rng.random(B) < 0.25).astype(np.float32)).to(dev)
    cross, head = DCNMixLayer(S, num_layer=L, num_expert=N), MultiDenseLayer(1, 1)
    head(cross(x[:256]))
    return DCNMixPairwiseStep(cross, head, x, y, g)


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = make(65536, 1)
print('full batch, one stream      : %.3f ms' % timed(full.run), flush=True)
h = [make(32768, 2), make(32768, 3)]
print('one half, one stream        : %.3f ms' % timed(h[0].run), flush=True)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def both():
    torch.cuda.set_stream(sa)
    h[0].run()
    torch.cuda.set_stream(sb)
    h[1].run()


print('two halves, two streams     : %.3f ms' % timed(both), flush=True)
q = [make(16384, 4 + i) for i in range(4)]
ss = [torch.cuda.Stream(dev) for _ in range(4)]


def four():
    for s, st in zip(ss, q):
        torch.cuda.set_stream(s)
        st.run()


print('four quarters, four streams : %.3f ms' % timed(four), flush=True)
