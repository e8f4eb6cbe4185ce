"""InnerPNNLayer -- drop-in for rec_now/layers/inner_pnn_layer.py (/root/reference/rec_now/layers/inner_pnn_layer.py:12-53)."""
import torch

from .. import _lib
from ._keras import Layer


class _InnerPNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *fields):
        xs = [_lib.f32c(x, 'InnerPNN input') for x in fields]
        B, D = xs[0].shape
        for x in xs:
            if x.shape != (B, D):
                raise ValueError('all InnerPNN inputs must have the same (B, D) shape')
        F = len(xs)
        if D > 64:
            raise NotImplementedError('InnerPNNLayer kernels cover embedding_dim <= 64 (the reference has no limit); got %d' % D)
        dev = xs[0].device
        out = torch.empty((B, F * (F - 1) // 2), dtype=torch.float32, device=dev)
        ptrs = _lib.ptr_array(xs, dev)
        _lib.call('recnow_inner_pnn_fwd', _lib.ptr(ptrs), F, B, D, _lib.ptr(out), _lib.stream())
        ctx.save_for_backward(*xs)
        return out

    @staticmethod
    def backward(ctx, dout):
        xs = ctx.saved_tensors
        F = len(xs)
        B, D = xs[0].shape
        dev = xs[0].device
        dout = _lib.f32c(dout, 'grad')
        dx = torch.empty((F, B, D), dtype=torch.float32, device=dev)
        ptrs = _lib.ptr_array(list(xs), dev)
        dptrs = _lib.block_ptr_array(dx, F)
        _lib.call('recnow_inner_pnn_bwd', _lib.ptr(ptrs), _lib.ptr(dptrs), F, B, D, _lib.ptr(dout), _lib.stream())
        return tuple(dx.unbind(0))


class InnerPNNLayer(Layer):
    """Inner Product-based Neural Network layer: the F(F-1)/2 pairwise inner products of the field embeddings.

    Symbols: B batch size, D embedding dim, F number of fields, P = C(F, 2).
    """

    def call(self, inputs):
        """inputs: list of F tensors of shape (B, D).  Returns (B, P), pairs (r, c), r < c, in r-major order (:41-45)."""
        return _InnerPNNFunction.apply(*inputs)
