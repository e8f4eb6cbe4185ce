"""attention_by_dot_product -- drop-in for rec_now/rec_block/attention.py:12-38
(/root/reference/rec_now/rec_block/attention.py).  One fused HIP kernel per direction (HBM-bound: the (B,L,D) user
embeddings are read once forward, once backward with the scores recomputed)."""
import torch

from .. import _lib


class _AttnDotFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user_emb, doc_emb, filter_neg):
        u = _lib.f32c(user_emb, 'user_emb')
        d = _lib.f32c(doc_emb, 'doc_emb')
        if u.dim() != 3 or d.dim() != 2 or u.shape[0] != d.shape[0] or u.shape[2] != d.shape[1]:
            raise ValueError('user_emb must be (B, L, D) and doc_emb (B, D); got %s and %s' % (tuple(u.shape), tuple(d.shape)))
        B, L, D = u.shape
        if D > 256:
            raise NotImplementedError('attention_by_dot_product kernels cover embedding_dim <= 256 (the reference has no limit); got %d' % D)
        mat = torch.empty((B, D), dtype=torch.float32, device=u.device)
        ssum = torch.empty((B, 1), dtype=torch.float32, device=u.device)
        _lib.call('recnow_attention_dot_fwd', _lib.ptr(u), _lib.ptr(d), B, L, D, 1 if filter_neg else 0, _lib.ptr(mat),
                  _lib.ptr(ssum), _lib.stream())
        ctx.save_for_backward(u, d)
        ctx.filter_neg = bool(filter_neg)
        return mat, ssum

    @staticmethod
    def backward(ctx, dmat, dsum):
        u, d = ctx.saved_tensors
        B, L, D = u.shape
        dmat = _lib.f32c(dmat, 'grad') if dmat is not None else None
        dsum = _lib.f32c(dsum, 'grad').reshape(-1) if dsum is not None else None
        du = torch.empty_like(u)
        dd = torch.empty_like(d)
        _lib.call('recnow_attention_dot_bwd', _lib.ptr(u), _lib.ptr(d), _lib.ptr(dmat), _lib.ptr(dsum), B, L, D,
                  1 if ctx.filter_neg else 0, _lib.ptr(du), _lib.ptr(dd), _lib.stream())
        return du, dd, None


def attention_by_dot_product(user_emb, doc_emb, filter_neg=False):
    """Dot-product attention of L user-feature embeddings against one item embedding.

    Args:
        user_emb: (B, L, D);  doc_emb: (B, D);  filter_neg: clamp negative scores to 0 (:31-32).
    Returns:
        attn_mat (B, D) = sum_l user_emb[:, l] * score_l,  attn_score_sum (B, 1) = sum_l score_l.
    """
    return _AttnDotFunction.apply(user_emb, doc_emb, filter_neg)
