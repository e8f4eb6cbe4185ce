"""Model-level fusion of the north-star step: DCNMixLayer -> MultiDenseLayer(1, 1) scoring head as ONE
autograd node over `recnow_dcn_mix_score_fwd/bwd` (include/recnow.h).

    scores = dcn_mix_score(cross, head, x)        # == head(cross(x)).reshape(-1), same weights, same gradients

replaces, for the composition /root/reference/rec_now/layers/dcn_mix_layer.py:114-151 -> multi_dense_layer.py:80-94 (units = 1,
num_dnn = 1, linear), three kernels of the head (forward dot, rank-one dy = dscore (x) w_head written to HBM, weight gradient),
the store of the last layer's output and the re-reads of dy by the top layer's backward products.  The two drop-in layers stay
what holds the weights (reference names, `named_weights()`); shapes the fused route does not cover fall back to calling them.

`layer_events`: optional list of L `GpuEvent`s; event l is recorded on the stream once every weight gradient of cross layer l has
been issued (the head's gradients belong to event L-1), so `dp.LayerwiseReducer` can all-reduce a layer's gradients while the
backward of the layers below still runs.
"""
import ctypes

import torch

from . import _lib
from .layers._ops import DCN_MIX_TWO_STREAMS, _host_ptr_array, pad_rows, ragged_pad_rows


class GpuEvent(object):
    """A HIP event owned through the C ABI (recnow_event_*): recorded by the library inside a backward pass, waited on by a
    side stream.  torch.cuda.Event creates its handle lazily at the first record, which is too late to hand to the library."""

    def __init__(self):
        h = ctypes.c_void_p()
        _lib.call('recnow_event_create', ctypes.byref(h))
        self.handle = h

    def wait(self, stream):
        """Make `stream` (torch.cuda.Stream) wait for the most recent record of this event."""
        _lib.call('recnow_stream_wait_event', ctypes.c_void_p(stream.cuda_stream), self.handle)

    def __del__(self):
        try:
            if self.handle:
                _lib.load().recnow_event_destroy(self.handle)
        except Exception:       # interpreter shutdown
            pass


class DCNMixScoreFunction(torch.autograd.Function):
    """params = U_0..U_{L-1}, V_0.., W_0.., bias_0.., gate_0.. (5*L tensors), as DCNMixFunction."""

    @staticmethod
    def forward(ctx, x, head_w, head_b, L, act_inner, act_outer, events, grad_buffers, *params):
        x = _lib.f32c(x, 'inputs')
        hw = _lib.f32c(head_w, 'head kernel').reshape(-1)
        hb = _lib.f32c(head_b, 'head bias').reshape(-1) if head_b is not None else None
        ps = [_lib.f32c(p, 'weight') for p in params]
        U, V, W, bias, gate = (ps[i * L:(i + 1) * L] for i in range(5))
        N, D, S = U[0].shape
        rows = x.shape[0]
        padded = ragged_pad_rows(rows, D, S, N, L)      # a ragged per-rank batch: zero-padded copy on the exact-128 route (layers/_ops.py)
        if padded:
            x = pad_rows(x, padded)
        B = x.shape[0]
        lib = _lib.load()
        saved = _lib.workspace(lib.recnow_dcn_mix_saved_bytes(B, D, S, N, L), x.device)
        ws = _lib.workspace(lib.recnow_dcn_mix_workspace_bytes(B, D, S, N, L), x.device)
        scores = torch.empty(B, dtype=torch.float32, device=x.device)
        need_dx = bool(ctx.needs_input_grad[0])
        _lib.call('recnow_dcn_mix_score_fwd', _lib.ptr(x), _host_ptr_array(U), _host_ptr_array(V), _host_ptr_array(W),
                  _host_ptr_array(bias), _host_ptr_array(gate), _lib.ptr(hw), _lib.ptr(hb), B, D, S, N, L, act_inner, act_outer,
                  _lib.ptr(scores), _lib.ptr(saved), saved.numel(), _lib.ptr(ws), ws.numel(), _lib.stream(), int(need_dx))
        ctx.save_for_backward(x, saved, hw, *ps)
        ctx.meta = (B, D, S, N, L, act_inner, act_outer, need_dx, head_w.shape, None if head_b is None else head_b.shape, events,
                    grad_buffers)
        ctx.rows = rows
        return scores[:rows] if padded else scores

    @staticmethod
    def backward(ctx, dscores):
        x, saved, hw, *ps = ctx.saved_tensors
        B, D, S, N, L, act_inner, act_outer, need_dx, hw_shape, hb_shape, events, gbuf = ctx.meta
        U, V, W, bias, gate = (ps[i * L:(i + 1) * L] for i in range(5))
        ds = _lib.f32c(dscores, 'grad').reshape(-1)
        if ctx.rows != B:
            ds = pad_rows(ds, B)
        dx = torch.empty_like(x) if need_dx else None
        # gradient storage: fresh tensors, or the caller's buffers (dp.LayerwiseReducer hands out views of one flat bucket per
        # layer, so the all-reduce runs in place: no packing before and no copy after the collective)
        out = lambda i, like: (gbuf[i].view(like.shape) if gbuf is not None and gbuf[i] is not None else torch.empty_like(like))    # noqa: E731
        grads = [out(2 + i, p) for i, p in enumerate(ps)]
        dU, dV, dW, dbias, dgate = (grads[i * L:(i + 1) * L] for i in range(5))
        dhw = out(0, hw)
        dhb = None
        if hb_shape is not None:
            dhb = gbuf[1].view(1) if gbuf is not None and gbuf[1] is not None else torch.empty(1, dtype=torch.float32, device=x.device)
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_dcn_mix_workspace_bytes(B, D, S, N, L), x.device)
        ev = None
        if events is not None:
            if len(events) != L:
                raise ValueError('layer_events must hold one event per cross layer')
            ev = (ctypes.c_void_p * L)(*[e.handle if e is not None else None for e in events])
        _lib.call('recnow_dcn_mix_score_bwd', _lib.ptr(x), _host_ptr_array(U), _host_ptr_array(V), _host_ptr_array(W),
                  _host_ptr_array(bias), _host_ptr_array(gate), _lib.ptr(hw), _lib.ptr(ds), _lib.ptr(saved), saved.numel(), B, D, S, N, L,
                  act_inner, act_outer, _lib.ptr(dx) if need_dx else None, _host_ptr_array(dU), _host_ptr_array(dV),
                  _host_ptr_array(dW), _host_ptr_array(dbias), _host_ptr_array(dgate), _lib.ptr(dhw), _lib.ptr(dhb), _lib.ptr(ws),
                  ws.numel(), _lib.stream(), _lib.side_stream(x.device) if DCN_MIX_TWO_STREAMS else None, ev)
        if need_dx and ctx.rows != B:
            dx = dx[:ctx.rows]
        return (dx, dhw.view(hw_shape), None if dhb is None else dhb.view(hb_shape), None, None, None, None, None) + tuple(grads)


def fused_route_available(cross, head, x, rows=None):
    """True when `dcn_mix_score` runs as one fused node for this input (else it calls the two layers).  rows: the row count to ask for
    instead of x.shape[0] (step.py: the padded storage of a ragged batch)."""
    if not (cross.built and head.built):
        return False
    if cross._cb_inner is not None or cross._cb_outer is not None:
        return False
    if head.num_dnn != 1 or head.units != 1 or head.act_callable is not None or head.act_code != 0:
        return False
    if x.dim() != 2 or not x.is_cuda:
        return False
    N, D, S = cross.origin_to_sub_kernels[0].shape
    if rows is None:
        rows = ragged_pad_rows(x.shape[0], D, S, N, cross.num_layer) or x.shape[0]      # a ragged batch takes the route on padded rows
    return bool(_lib.load().recnow_dcn_mix_score_supported(rows, D, S, N, cross.num_layer))


def score_params(cross, head):
    """The node's parameters in the order of its gradient buffers: head kernel, head bias, then U_0.., V_0.., W_0.., bias_0.., gate_0.."""
    return ([head.kernel, head.bias] + list(cross.origin_to_sub_kernels) + list(cross.sub_to_sub_kernels)
            + list(cross.sub_to_origin_kernels) + list(cross.biases) + [g.kernel for g in cross.gate_layers])


def dcn_mix_score(cross, head, x, layer_events=None, grad_buffers=None):
    """scores (B,) = head(cross(x)).reshape(-1) for a DCNMixLayer `cross` and a MultiDenseLayer(1, 1) `head` (linear).
    grad_buffers: optional list aligned with `score_params(cross, head)` of preallocated gradient tensors (or None entries)."""
    if not (cross.built and head.built):
        return head(cross(x)).reshape(-1)            # builds both (Keras-style lazy build); the next call takes the fused route
    if not fused_route_available(cross, head, x):
        return head(cross(x)).reshape(-1)
    params = (list(cross.origin_to_sub_kernels) + list(cross.sub_to_sub_kernels) + list(cross.sub_to_origin_kernels)
              + list(cross.biases) + [g.kernel for g in cross.gate_layers])
    if grad_buffers is not None:
        # A parameter whose .grad still IS its buffer (gradient accumulation: a second backward without `p.grad = None` in between)
        # must not have that storage overwritten by this pass before autograd adds the new gradient to it: such a parameter gets a
        # fresh gradient tensor (autograd then accumulates into the bucket view, dp.LayerwiseReducer.reduce sees it as in place).
        owners = [head.kernel, head.bias] + params
        grad_buffers = [None if (b is not None and p is not None and p.grad is not None and p.grad.data_ptr() == b.data_ptr()) else b
                        for p, b in zip(owners, grad_buffers)]
    return DCNMixScoreFunction.apply(x, head.kernel, head.bias, cross.num_layer, cross._act_inner, cross._act_outer, layer_events,
                                     grad_buffers, *params)
