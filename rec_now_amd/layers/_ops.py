"""torch.autograd.Function wrappers over the C ABI for the layer kernels (host plumbing only: shapes, buffers, streams)."""
import ctypes
import os

import torch

from .. import _lib


def _host_ptr_array(tensors):
    """HOST array of device pointers (the `*_host` parameters of include/recnow.h)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


# ---- MultiDense ----------------------------------------------------------------------------------------------
class MultiDenseFunction(torch.autograd.Function):
    """y[n] = act(x[n] @ kernel[n] + bias[n]); x (B,D) broadcast or (N,B,D); kernel (N,D,U); bias (N,1,U) or None."""

    @staticmethod
    def forward(ctx, x, kernel, bias, act_code):
        x = _lib.f32c(x, 'inputs')
        kernel = _lib.f32c(kernel, 'kernel')
        N, D, U = kernel.shape
        batched = x.dim() == 3
        B = x.shape[-2]
        if bias is not None:
            bias = _lib.f32c(bias, 'bias')
        y = torch.empty((N, B, U), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_multi_dense_workspace_bytes(B, D, U, N), x.device)
        _lib.call('recnow_multi_dense_fwd', _lib.ptr(x), 1 if batched else 0, _lib.ptr(kernel), _lib.ptr(bias), B, D, U, N,
                  act_code, _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.save_for_backward(x, kernel, y)
        ctx.meta = (batched, B, D, U, N, act_code, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, kernel, y = ctx.saved_tensors
        batched, B, D, U, N, act_code, has_bias = ctx.meta
        dy = _lib.f32c(dy, 'grad')
        need_x, need_k, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dk = torch.empty_like(kernel) if need_k else None
        db = torch.empty((N, 1, U), dtype=torch.float32, device=x.device) if need_b else None
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_multi_dense_workspace_bytes(B, D, U, N), x.device)
        _lib.call('recnow_multi_dense_bwd', _lib.ptr(x), 1 if batched else 0, _lib.ptr(kernel), _lib.ptr(y), _lib.ptr(dy), B, D,
                  U, N, act_code, _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.stream())
        return dx, dk, db, None


def multi_dense(x, kernel, bias, act_code):
    return MultiDenseFunction.apply(x, kernel, bias, act_code)


# ---- gate softmax + expert mixing ------------------------------------------------------------------------------
class MoeMixFunction(torch.autograd.Function):
    """out[t] = sum_n softmax(logits[t])[:, n, None] * experts[n];  logits (T,B,N); experts: N tensors (B,U)."""

    @staticmethod
    def forward(ctx, logits, *experts):
        logits = _lib.f32c(logits, 'gate logits')
        T, B, N = logits.shape
        es = [_lib.f32c(e, 'expert output') for e in experts]
        if len(es) != N:
            raise ValueError('gate has %d outputs but %d experts were given' % (N, len(es)))
        U = es[0].shape[-1]
        dev = logits.device
        gates = torch.empty((T, B, N), dtype=torch.float32, device=dev)
        out = torch.empty((T, B, U), dtype=torch.float32, device=dev)
        ptrs = _lib.ptr_array(es, dev)
        _lib.call('recnow_moe_mix_fwd', _lib.ptr(logits), _lib.ptr(ptrs), T, B, N, U, _lib.ptr(gates), _lib.ptr(out), _lib.stream())
        ctx.save_for_backward(gates, *es)
        ctx.meta = (T, B, N, U)
        return out

    @staticmethod
    def backward(ctx, dout):
        gates, *es = ctx.saved_tensors
        T, B, N, U = ctx.meta
        dev = gates.device
        dout = _lib.f32c(dout, 'grad')
        dlogits = torch.empty_like(gates)
        des = torch.empty((N, B, U), dtype=torch.float32, device=dev)
        ptrs = _lib.ptr_array(es, dev)
        dptrs = _lib.ptr_array([des[n] for n in range(N)], dev)
        _lib.call('recnow_moe_mix_bwd', _lib.ptr(gates), _lib.ptr(ptrs), _lib.ptr(dout), T, B, N, U, _lib.ptr(dlogits),
                  _lib.ptr(dptrs), 0, _lib.stream())
        return (dlogits,) + tuple(des.unbind(0))


def moe_mix(logits, experts):
    """logits (T,B,N) or (B,N); experts: list of N (B,U) tensors.  Returns (T,B,U) or (B,U)."""
    squeeze = logits.dim() == 2
    if squeeze:
        logits = logits.unsqueeze(0)
    out = MoeMixFunction.apply(logits, *experts)
    return out[0] if squeeze else out


# ---- DCN-v1 ----------------------------------------------------------------------------------------------------
class DCNFunction(torch.autograd.Function):
    """All L cross layers in one call.  kernels (L,D), biases (L,D) or None.  Any L and D: the library picks the fused register
    kernels (L <= 4, D <= 4096) or the general streaming kernels."""

    @staticmethod
    def forward(ctx, x, kernels, biases, act_code):
        x = _lib.f32c(x, 'inputs')
        kernels = _lib.f32c(kernels, 'kernels')
        if biases is not None:
            biases = _lib.f32c(biases, 'biases')
        B, D = x.shape
        L = kernels.shape[0]
        y = torch.empty_like(x)
        csave = torch.empty((B, L), dtype=torch.float32, device=x.device)      # per-row scalars x_l . w_l for the backward
        _lib.call('recnow_dcn_fwd', _lib.ptr(x), _lib.ptr(kernels), _lib.ptr(biases), B, D, L, act_code, _lib.ptr(y), _lib.ptr(csave),
                  _lib.stream())
        ctx.save_for_backward(x, kernels, biases if biases is not None else x.new_empty(0), csave)
        ctx.meta = (B, D, L, act_code, biases is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, kernels, biases, csave = ctx.saved_tensors
        B, D, L, act_code, has_bias = ctx.meta
        dy = _lib.f32c(dy, 'grad')
        dx = torch.empty_like(x)
        dk = torch.empty_like(kernels)
        db = torch.empty_like(kernels) if has_bias else None
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_dcn_workspace_bytes(B, D, L), x.device)
        _lib.call('recnow_dcn_bwd', _lib.ptr(x), _lib.ptr(kernels), _lib.ptr(biases) if has_bias else None, _lib.ptr(dy), _lib.ptr(csave),
                  B, D, L, act_code, _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.stream())
        return dx, dk, db, None


class DCNStepFunction(torch.autograd.Function):
    """One cross layer with its own layer input: z = act(x0 * (x_l . w) + b); w (D,), b (D,) or None.  The route for
    activations that are user callables (act_code 0 here, the callable runs on z)."""

    @staticmethod
    def forward(ctx, x0, xl, w, b, act_code):
        x0 = _lib.f32c(x0, 'inputs')
        xl = _lib.f32c(xl, 'layer input')
        w = _lib.f32c(w, 'kernel').reshape(-1)
        if b is not None:
            b = _lib.f32c(b, 'bias').reshape(-1)
        B, D = x0.shape
        z = torch.empty_like(x0)
        c = torch.empty(max(B, 1), dtype=torch.float32, device=x0.device)
        _lib.call('recnow_dcn_step_fwd', _lib.ptr(x0), _lib.ptr(xl), _lib.ptr(w), _lib.ptr(b), B, D, act_code, _lib.ptr(z), _lib.ptr(c),
                  _lib.stream())
        ctx.save_for_backward(x0, xl, w, z if act_code else x0.new_empty(0), c)
        ctx.meta = (B, D, act_code, b is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        x0, xl, w, z, c = ctx.saved_tensors
        B, D, act_code, has_bias = ctx.meta
        dz = _lib.f32c(dz, 'grad')
        dx0, dxl = torch.empty_like(x0), torch.empty_like(x0)
        dw = torch.empty(D, dtype=torch.float32, device=x0.device)
        db = torch.empty(D, dtype=torch.float32, device=x0.device) if has_bias else None
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_dcn_step_workspace_bytes(B, D), x0.device)
        _lib.call('recnow_dcn_step_bwd', _lib.ptr(x0), _lib.ptr(xl), _lib.ptr(w), _lib.ptr(z) if act_code else None, _lib.ptr(c),
                  _lib.ptr(dz), B, D, act_code, _lib.ptr(dx0), _lib.ptr(dxl), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), ws.numel(),
                  _lib.stream())
        return dx0, dxl, dw, db, None


# Opt-in: run the weight-gradient products of the DCN-v2 backward on a second HIP stream (recnow_dcn_mix_bwd's stream2).
# Measured on MI355X at the north-star shape: 5.43 vs 5.54 ms/step (-2 %), every GEMM already fills all 256 CUs, so the
# default stays single-stream (per-launch timings then remain meaningful for the roofline hook).
DCN_MIX_TWO_STREAMS = os.environ.get('RECNOW_TWO_STREAMS', '0') in ('1', '2')


# ---- DCN-v2 mix ---------------------------------------------------------------------------------------------------
def ragged_pad_rows(B, D, S, N, L):
    """Rows of padded storage for a RAGGED batch on the exact-128 routes, or 0 (run as it is).

    The per-rank batches of data parallelism are whole groups (dp.shard_rows_by_group), i.e. not multiples of 256, and the fast
    routes of the cross layers (exact-128 formulation, fused sub-space kernels, row-block kernels) want whole 256-row blocks.  A
    batch of at least RECNOW_PAD_MIN_ROWS rows (default 2048; smaller ones stay on the general kernels, where the padding would be
    a large share of the work) whose shape the fast routes take once padded runs on a zero-padded copy: zero rows of x with a zero
    upstream gradient add exactly 0.0f to every weight gradient, and their outputs are sliced away.  RECNOW_PAD_RAGGED=0: off."""
    if B % 256 == 0 or os.environ.get('RECNOW_PAD_RAGGED') == '0' or B < int(os.environ.get('RECNOW_PAD_MIN_ROWS', '2048')):
        return 0
    Bp = -(-B // 256) * 256
    return Bp if _lib.load().recnow_dcn_mix_score_supported(Bp, D, S, N, L) else 0


def pad_rows(t, rows):
    """t (B, ...) -> (rows, ...) with zero rows behind the B rows of t."""
    out = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:t.shape[0]].copy_(t)
    return out


class DCNMixFunction(torch.autograd.Function):
    """All L layers of DCNMixLayer.  params = U_0..U_{L-1}, V_0.., W_0.., bias_0.., gate_0.. (5*L tensors)."""

    @staticmethod
    def forward(ctx, x, L, act_inner, act_outer, *params):
        x = _lib.f32c(x, 'inputs')
        ps = [_lib.f32c(p, 'weight') for p in params]
        U, V, W, bias, gate = (ps[i * L:(i + 1) * L] for i in range(5))
        N, D, S = U[0].shape
        rows = x.shape[0]
        padded = ragged_pad_rows(rows, D, S, N, L)
        if padded:
            x = pad_rows(x, padded)
        B = x.shape[0]
        lib = _lib.load()
        saved = _lib.workspace(lib.recnow_dcn_mix_saved_bytes(B, D, S, N, L), x.device)
        ws = _lib.workspace(lib.recnow_dcn_mix_workspace_bytes(B, D, S, N, L), x.device)
        y = torch.empty_like(x)
        # x is data (no gradient asked for, as under tf.GradientTape.gradient(loss, weights)): nothing that only dx needs
        # is kept by the forward or launched by the backward
        need_dx = bool(ctx.needs_input_grad[0])
        _lib.call('recnow_dcn_mix_fwd', _lib.ptr(x), _host_ptr_array(U), _host_ptr_array(V), _host_ptr_array(W),
                  _host_ptr_array(bias), _host_ptr_array(gate), B, D, S, N, L, act_inner, act_outer, _lib.ptr(y),
                  _lib.ptr(saved), saved.numel(), _lib.ptr(ws), ws.numel(), _lib.stream(), int(need_dx))
        ctx.save_for_backward(x, saved, *ps)
        ctx.meta = (B, D, S, N, L, act_inner, act_outer, need_dx)
        ctx.rows = rows
        return y[:rows] if padded else y

    @staticmethod
    def backward(ctx, dy):
        x, saved, *ps = ctx.saved_tensors
        B, D, S, N, L, act_inner, act_outer, need_dx = ctx.meta
        U, V, W, bias, gate = (ps[i * L:(i + 1) * L] for i in range(5))
        dy = _lib.f32c(dy, 'grad')
        if ctx.rows != B:
            dy = pad_rows(dy, B)
        dx = torch.empty_like(x) if need_dx else None
        grads = [torch.empty_like(p) for p in ps]
        dU, dV, dW, dbias, dgate = (grads[i * L:(i + 1) * L] for i in range(5))
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_dcn_mix_workspace_bytes(B, D, S, N, L), x.device)
        _lib.call('recnow_dcn_mix_bwd', _lib.ptr(x), _host_ptr_array(U), _host_ptr_array(V), _host_ptr_array(W),
                  _host_ptr_array(bias), _host_ptr_array(gate), _lib.ptr(dy), _lib.ptr(saved), saved.numel(), B, D, S, N, L,
                  act_inner, act_outer, _lib.ptr(dx) if need_dx else None, _host_ptr_array(dU), _host_ptr_array(dV), _host_ptr_array(dW),
                  _host_ptr_array(dbias), _host_ptr_array(dgate), _lib.ptr(ws), ws.numel(), _lib.stream(),
                  _lib.side_stream(x.device) if DCN_MIX_TWO_STREAMS else None)
        if need_dx and ctx.rows != B:
            dx = dx[:ctx.rows]
        return (dx, None, None, None) + tuple(grads)


# ---- CIN --------------------------------------------------------------------------------------------------------------
class CINFunction(torch.autograd.Function):
    """emb (B, F*D); weights: L tensors, W_k viewed (H_k, F*H_{k-1})."""

    @staticmethod
    def forward(ctx, emb, D, F, hidden, output_input, sum_channel, *weights):
        emb = _lib.f32c(emb, 'inputs')
        ws_ = [_lib.f32c(w, 'weight') for w in weights]
        B = emb.shape[0]
        L = len(hidden)
        hid = (ctypes.c_int * L)(*hidden)
        lib = _lib.load()
        saved = _lib.workspace(lib.recnow_cin_saved_bytes(B, D, F, hid, L), emb.device)
        ws = _lib.workspace(lib.recnow_cin_workspace_bytes(B, D, F, hid, L), emb.device)
        ctot = (F if output_input else 0) + sum(hidden)
        out = torch.empty((B, D if sum_channel else ctot * D), dtype=torch.float32, device=emb.device)
        _lib.call('recnow_cin_fwd', _lib.ptr(emb), _host_ptr_array(ws_), B, D, F, hid, L, 1 if output_input else 0,
                  1 if sum_channel else 0, _lib.ptr(out), _lib.ptr(saved), saved.numel(), _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.save_for_backward(saved, *ws_)
        ctx.meta = (B, D, F, tuple(hidden), output_input, sum_channel)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, *ws_ = ctx.saved_tensors
        B, D, F, hidden, output_input, sum_channel = ctx.meta
        L = len(hidden)
        hid = (ctypes.c_int * L)(*hidden)
        dout = _lib.f32c(dout, 'grad')
        demb = torch.empty((B, F * D), dtype=torch.float32, device=dout.device)
        dws = [torch.empty_like(w) for w in ws_]
        lib = _lib.load()
        ws = _lib.workspace(lib.recnow_cin_workspace_bytes(B, D, F, hid, L), dout.device)
        _lib.call('recnow_cin_bwd', _host_ptr_array(ws_), _lib.ptr(dout), _lib.ptr(saved), saved.numel(), B, D, F, hid, L,
                  1 if output_input else 0, 1 if sum_channel else 0, _lib.ptr(demb), _host_ptr_array(dws), _lib.ptr(ws),
                  ws.numel(), _lib.stream())
        return (demb, None, None, None, None, None) + tuple(dws)
