"""TEST INFRASTRUCTURE: the fp64 oracle (oracle/dense_ref.py) evaluated over a large batch in row chunks.

The interaction layers are row-separable: outputs and input gradients of a row depend on that row only, and every weight
gradient is a plain sum over rows.  So the oracle at BASELINE.json's sizes (B = 65536 x D = 1024, where one fp64 autograd
graph of the whole batch would need tens of GB) is the chunk-wise oracle with the weight gradients accumulated in fp64."""
import os

import numpy as np
import torch


GEMM_PRECISIONS = ('f32', 'bf16x3')      # the two arithmetics of the DCN-v2 products: every north-star parity test holds BOTH to the same 1e-5 bound


class gemm_precision:
    """`with gemm_precision('bf16x3'):` -- the split-precision products (recnow_set_gemm_precision(1): fp32 operands as three bf16 pieces, six bf16
    MFMA terms, fp32 accumulation) for the block, the exact-fp32 kernels restored behind it."""
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from rec_now_amd import _lib
        _lib.call('recnow_set_gemm_precision', 1 if self.mode == 'bf16x3' else 0)
        return self.mode

    def __exit__(self, *exc):
        from rec_now_amd import _lib
        _lib.call('recnow_set_gemm_precision', 0)
        return False


def weights64(named):
    """{name: tensor} -> {name: fp64 CPU leaf with requires_grad}."""
    return {k: v.detach().cpu().double().requires_grad_(True) for k, v in named.items()}


def run_chunked(fwd, x, gy, weights, chunk=4096, want_dx=True):
    """fwd(x64_chunk) -> output tensor or list of tensors, built from the fp64 leaves in `weights` (a dict).
    gy: upstream gradient(s), same structure as the output, or a callable (lo, hi, outs) -> list of gradients.
    Returns (outs, dx, wgrads): outs = list of np.float64 arrays for the full batch, dx np array (or None), wgrads dict."""
    B = x.shape[0]
    outs_all, dx_all = None, (np.empty(tuple(x.shape), np.float64) if want_dx else None)
    acc = {k: torch.zeros_like(v) for k, v in weights.items()}
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        for v in weights.values():
            v.grad = None
        xc = x[lo:hi].detach().cpu().double().requires_grad_(want_dx and gy is not None)
        if gy is None:                 # forward only (the scores pass in front of a loss): no autograd graph to build
            with torch.no_grad():
                out = fwd(xc)
        else:
            out = fwd(xc)
        outs = list(out) if isinstance(out, (list, tuple)) else [out]
        if outs_all is None:
            outs_all = [np.empty((B,) + tuple(o.shape[1:]), np.float64) for o in outs]
        for dst, o in zip(outs_all, outs):
            dst[lo:hi] = o.detach().numpy()
        if gy is not None:
            if callable(gy):
                gs = gy(lo, hi, outs)
            else:
                gs = [g[lo:hi].detach().cpu().double() for g in (gy if isinstance(gy, (list, tuple)) else [gy])]
            torch.autograd.backward(outs, gs)
            if want_dx:
                dx_all[lo:hi] = xc.grad.numpy()
            for k, v in weights.items():
                if v.grad is not None:
                    acc[k] += v.grad
    return outs_all, dx_all, {k: v.numpy() for k, v in acc.items()}


def close(a, b, rtol=1e-5, scale=None, what=''):
    a = a.detach().cpu().double().numpy() if hasattr(a, 'detach') else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if hasattr(b, 'detach') else np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    s = max(np.abs(b).max() if scale is None else scale, 1e-30)
    err = np.abs(a - b).max()
    if os.environ.get('RECNOW_TEST_MARGIN_LOG') and err > 0.3 * rtol * s:      # diagnostics: comparisons that use more than 30 % of their bound
        with open(os.environ['RECNOW_TEST_MARGIN_LOG'], 'a') as fh:
            fh.write('%.3f of the bound  %s  %s\n' % (err / (rtol * s), os.environ.get('PYTEST_CURRENT_TEST', ''), what))
    assert err <= rtol * s, '%s: max err %.3g vs scale %.3g (rel %.3g > %.1g)' % (what, err, s, err / s, rtol)
