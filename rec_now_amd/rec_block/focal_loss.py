"""focal_crossentropy_loss -- drop-in for rec_now/rec_block/focal_loss.py:12-66
(/root/reference/rec_now/rec_block/focal_loss.py).  One elementwise HIP kernel per direction; the mean is reduced in a
fixed order (double block partials), no float atomics."""
import torch

from .. import _lib


class _FocalFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, labels, logits, alpha, gamma, stop_weight_gradient, return_mean):
        shape = logits.shape
        z = _lib.f32c(labels, 'labels').reshape(-1)
        x = _lib.f32c(logits, 'logits').reshape(-1)
        if z.numel() != x.numel():
            raise ValueError('labels and logits must have the same number of elements')
        B = x.numel()
        dev = x.device
        lib = _lib.load()
        if return_mean:
            out = torch.empty((), dtype=torch.float32, device=dev)
            ws = _lib.workspace(lib.recnow_focal_loss_workspace_bytes(B), dev)
            _lib.call('recnow_focal_loss_fwd', _lib.ptr(z), _lib.ptr(x), B, alpha, gamma, None, _lib.ptr(out), _lib.ptr(ws),
                      ws.numel(), _lib.stream())
        else:
            out = torch.empty(shape, dtype=torch.float32, device=dev)
            _lib.call('recnow_focal_loss_fwd', _lib.ptr(z), _lib.ptr(x), B, alpha, gamma, _lib.ptr(out), None, None, 0,
                      _lib.stream())
        ctx.save_for_backward(z, x)
        ctx.meta = (alpha, gamma, bool(stop_weight_gradient), bool(return_mean), shape)
        return out

    @staticmethod
    def backward(ctx, g):
        z, x = ctx.saved_tensors
        alpha, gamma, stop_w, mean, shape = ctx.meta
        B = x.numel()
        g = _lib.f32c(g, 'grad')
        dx = torch.empty_like(x)
        if mean:
            _lib.call('recnow_focal_loss_bwd', _lib.ptr(z), _lib.ptr(x), B, alpha, gamma, 1 if stop_w else 0, None, _lib.ptr(g),
                      1.0 / max(B, 1), _lib.ptr(dx), _lib.stream())
        else:
            _lib.call('recnow_focal_loss_bwd', _lib.ptr(z), _lib.ptr(x), B, alpha, gamma, 1 if stop_w else 0,
                      _lib.ptr(g.reshape(-1)), None, 1.0, _lib.ptr(dx), _lib.stream())
        return None, dx.reshape(shape), None, None, None, None


def focal_crossentropy_loss(labels, logits, alpha=0.25, gamma=2.0, stop_weight_gradient=False, return_mean=True):
    """Focal loss (https://arxiv.org/pdf/1708.02002.pdf) on logits.

    Args:
        labels, logits: same shape, e.g. (B,) or (B, 1); labels in {0, 1}.
        alpha: weight of positives (1 - alpha for negatives); None/0 disables (:50).
        gamma: focusing exponent of (1 - p_t); None/0 disables (:55).
        stop_weight_gradient: no gradient through the modulating factor (:60-61).
        return_mean: scalar mean if True, else per-element losses with the shape of logits.
    Raises:
        ValueError: alpha outside (0, 1) or gamma < 0 (:43-46).
    """
    if alpha and (alpha <= 0.0 or alpha >= 1.0):
        raise ValueError('Value of alpha should be greater than zero and less than one.')
    if gamma and gamma < 0:
        raise ValueError('Value of gamma should be greater than or equal to zero.')
    return _FocalFunction.apply(labels, logits, float(alpha) if alpha else 0.0, float(gamma) if gamma else 0.0,
                                stop_weight_gradient, return_mean)
