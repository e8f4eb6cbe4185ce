"""FMLayer -- drop-in for rec_now/layers/fm_layer.py (/root/reference/rec_now/layers/fm_layer.py:12-42)."""
import torch

from .. import _lib
from ._keras import Layer


class _FMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *fields):
        xs = [_lib.f32c(x, 'FM input') for x in fields]
        B, D = xs[0].shape
        for x in xs:
            if x.shape != (B, D):
                raise ValueError('all FM inputs must have the same (B, D) shape')
        dev = xs[0].device
        y = torch.empty((B, 1), dtype=torch.float32, device=dev)
        S = torch.empty((B, D), dtype=torch.float32, device=dev)
        ptrs = _lib.ptr_array(xs, dev)
        _lib.call('recnow_fm_fwd', _lib.ptr(ptrs), len(xs), B, D, _lib.ptr(y), _lib.ptr(S), _lib.stream())
        ctx.save_for_backward(S, *xs)
        return y

    @staticmethod
    def backward(ctx, gy):
        S, *xs = ctx.saved_tensors
        B, D = S.shape
        F = len(xs)
        gy = _lib.f32c(gy, 'grad').reshape(-1)
        dx = torch.empty((F, B, D), dtype=torch.float32, device=S.device)
        ptrs = _lib.ptr_array(xs, S.device)
        dptrs = _lib.block_ptr_array(dx, F)
        _lib.call('recnow_fm_bwd', _lib.ptr(ptrs), _lib.ptr(dptrs), F, B, D, _lib.ptr(S), _lib.ptr(gy), _lib.stream())
        return tuple(dx.unbind(0))


class FMLayer(Layer):
    """Second-order Factorization-Machine interaction.

    Symbols: B batch size, D embedding dim, F number of fields.
    """

    def call(self, inputs):
        """inputs: list of F tensors of shape (B, D) (a single tensor is wrapped into a 1-element list, as the
        reference does at fm_layer.py:33-34).  Returns (B, 1)."""
        if not isinstance(inputs, (list, tuple)):
            inputs = [inputs]
        return _FMFunction.apply(*inputs)
