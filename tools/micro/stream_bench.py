"""Kernel-only timings (HIP events, 50 launches) of the streaming kernels whose bound is the HBM read rate:
FM forward / backward at BASELINE config 4, DCNLayer forward / backward at config 3's shape and the scoring head of the north-star step.  GPU only.
usage: python tools/micro/stream_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rec_now_amd import _lib

dev = torch.device('cuda:0')
lib = _lib.load()


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


def fm():
    B, F, D = 131072, 64, 16
    xs = [torch.randn(B, D, device=dev) for _ in range(F)]
    y = torch.empty(B, 1, device=dev)
    S = torch.empty(B, D, device=dev)
    gy = torch.randn(B, 1, device=dev)
    dx = torch.empty(F, B, D, device=dev)
    ptrs = _lib.ptr_array(xs, dev)
    dptrs = _lib.block_ptr_array(dx, F)
    st = _lib.stream()
    tf = timeit(lambda: _lib.call('recnow_fm_fwd', _lib.ptr(ptrs), F, B, D, _lib.ptr(y), _lib.ptr(S), st))
    tb = timeit(lambda: _lib.call('recnow_fm_bwd', _lib.ptr(ptrs), _lib.ptr(dptrs), F, B, D, _lib.ptr(S), _lib.ptr(gy), st))
    nb = 4.0 * B * F * D
    print('FM fwd  B=%d F=%d D=%d : %.1f us  %.2f TB/s of 4*B*F*D bytes' % (B, F, D, tf, nb / tf / 1e6))
    print('FM bwd  B=%d F=%d D=%d : %.1f us  %.2f TB/s of 8*B*F*D bytes' % (B, F, D, tb, 2 * nb / tb / 1e6))


def head():
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    B, D = 65536, 1024
    x = torch.randn(B, D, device=dev, requires_grad=True)
    layer = MultiDenseLayer(1, 1)
    layer(x[:8])
    t = timeit(lambda: layer(x.detach()))
    print('head fwd  B=%d D=%d : %.1f us wall per call (kernel + host)  %.2f TB/s of 4*B*D bytes' % (B, D, t, 4.0 * B * D / t / 1e6))
    gy = torch.randn(B, device=dev)

    def step():
        x.grad = None
        layer(x).reshape(-1).backward(gy)
    t = timeit(step)
    print('head fwd+bwd : %.1f us wall per step  %.2f TB/s of 12*B*D bytes (x | x, dx)' % (t, 12.0 * B * D / t / 1e6))


def dcn():
    B, D, L = 65536, 1024, 3                 # BASELINE config 3 shape, DCNLayer (rec_now/layers/dcn_layer.py:91-103)
    x = torch.randn(B, D, device=dev)
    k = torch.randn(L, D, device=dev) * 0.03
    b = torch.randn(L, D, device=dev) * 0.03
    y, dy, dx = torch.empty_like(x), torch.randn(B, D, device=dev), torch.empty_like(x)
    cs = torch.empty(B, L, device=dev)
    dk, db = torch.empty_like(k), torch.empty_like(b)
    ws = _lib.workspace(lib.recnow_dcn_workspace_bytes(B, D, L), dev)
    st = _lib.stream()
    tf = timeit(lambda: _lib.call('recnow_dcn_fwd', _lib.ptr(x), _lib.ptr(k), _lib.ptr(b), B, D, L, 0, _lib.ptr(y), _lib.ptr(cs), st))
    tb = timeit(lambda: _lib.call('recnow_dcn_bwd', _lib.ptr(x), _lib.ptr(k), _lib.ptr(b), _lib.ptr(dy), _lib.ptr(cs), B, D, L, 0, _lib.ptr(dx),
                                  _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(), st))
    print('DCN fwd  B=%d D=%d L=%d : %.1f us  %.2f TB/s of 8*B*D bytes' % (B, D, L, tf, 8.0 * B * D / tf / 1e6))
    print('DCN bwd  B=%d D=%d L=%d : %.1f us  %.2f TB/s of 12*B*D bytes (all launches of the call)' % (B, D, L, tb, 12.0 * B * D / tb / 1e6))


if __name__ == '__main__':
    dcn()
    fm()
    head()
