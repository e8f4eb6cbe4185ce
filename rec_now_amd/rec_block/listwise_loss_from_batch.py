"""In-batch listwise samples / loss -- drop-in for rec_now/rec_block/listwise_loss_from_batch.py
(/root/reference/rec_now/rec_block/listwise_loss_from_batch.py).  Same public names, arguments, defaults and return
structure, torch.Tensor in place of tf.Tensor.

`to_listwise_sample` + `listwise_loss_via_softmax_cross_entropy_with_logits` return/consume the reference's dense
(num_valid_group, batch_size) matrices (materialised by a HIP scatter kernel, loss by a row-softmax kernel).
`listwise_loss_from_batch` is the same mathematics fused on sorted segments -- no (G,B) matrix, no host sync --
and is what a training step should call.
"""
import torch

from .. import _lib
from ._segments import build_segments


def _as_f32(x):
    return x if x.dtype == torch.float32 else x.to(torch.float32)


def row_not_all_zero(x):
    """(M,N) -> (M,) bool: the row has a non-zero element (reference :13-31)."""
    _lib.require_gpu(x, 'x')
    return (_as_f32(x) != 0.0).to(torch.int32).sum(dim=-1) > 0


def row_has_value_greater_than(x, threshold):
    """(M,N) -> (M,) bool: the row has an element > threshold (reference :34-53)."""
    _lib.require_gpu(x, 'x')
    return (_as_f32(x) > threshold).to(torch.int32).sum(dim=-1) > 0


def row_has_value_less_than(x, threshold):
    """(M,N) -> (M,) bool: the row has an element < threshold (reference :56-71)."""
    _lib.require_gpu(x, 'x')
    return (_as_f32(x) < threshold).to(torch.int32).sum(dim=-1) > 0


def nan_to_zero(val):
    """Scalar tensor: NaN -> 0, anything else unchanged; non-scalars raise ValueError (reference :74-86)."""
    if val.dim() != 0:
        raise ValueError('input muust be a scalar tf.Tensor')
    return torch.where(torch.isnan(val), torch.zeros((), dtype=val.dtype, device=val.device), val)


class _ListStats(object):
    __slots__ = ('seg', 'B', 'labels', 'logits', 'seg_valid', 'seg_lse', 'seg_ysum', 'seg_psum', 'seg_pdot', 'valid_rank',
                 'n_valid', 'pad_logit')


def _list_stats(group_ids, labels, logits, do_mask_logits, value_of_masked_logit, pos_neg_th):
    seg = build_segments(group_ids.reshape(-1) if isinstance(group_ids, torch.Tensor) else group_ids)
    B, dev = seg.B, seg.device
    st = _ListStats()
    st.seg, st.B = seg, B
    st.labels = _lib.f32c(labels, 'labels').reshape(-1).detach()
    st.logits = _lib.f32c(logits, 'logits').reshape(-1).detach()
    if st.labels.numel() != B or st.logits.numel() != B:
        raise ValueError('group_ids, labels and logits must have the same number of elements')
    n = max(B, 1)
    st.seg_valid = torch.empty(n, dtype=torch.int32, device=dev)
    st.seg_lse, st.seg_ysum, st.seg_psum, st.seg_pdot = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
    st.valid_rank = torch.empty(n, dtype=torch.int32, device=dev)
    st.n_valid = torch.empty(1, dtype=torch.int32, device=dev)
    st.pad_logit = float(value_of_masked_logit) if do_mask_logits else 0.0
    ws = _lib.workspace(_lib.load().recnow_listwise_workspace_bytes(B), dev)
    _lib.call('recnow_listwise_segments', _lib.ptr(st.labels), _lib.ptr(st.logits), _lib.ptr(seg.order), _lib.ptr(seg.seg_first),
              _lib.ptr(seg.n_seg), B, float(pos_neg_th), st.pad_logit, _lib.ptr(st.seg_valid), _lib.ptr(st.seg_lse),
              _lib.ptr(st.seg_ysum), _lib.ptr(st.seg_psum), _lib.ptr(st.seg_pdot), _lib.ptr(st.valid_rank), _lib.ptr(st.n_valid),
              _lib.ptr(ws), ws.numel(), _lib.stream())
    return st


def _row_rank(st):
    """valid rank of every ROW's group (or -1), via the fused kernel's by-product."""
    dev = st.seg.device
    n = max(st.B, 1)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dbase = torch.empty(n, dtype=torch.float32, device=dev)
    row_rank = torch.empty(n, dtype=torch.int32, device=dev)
    group_loss = torch.empty(n, dtype=torch.float32, device=dev)
    return loss, dbase, row_rank, group_loss


class _DenseLogits(torch.autograd.Function):
    """dense_logits (Gv,B) as a differentiable function of logits (B,)."""

    @staticmethod
    def forward(ctx, logits, dense_logits, row_rank, B):
        ctx.save_for_backward(row_rank)
        ctx.B = B
        ctx.shape = logits.shape
        return dense_logits

    @staticmethod
    def backward(ctx, g):
        (row_rank,) = ctx.saved_tensors
        B = ctx.B
        if g.numel() == 0:                 # no valid list: nothing flows back
            return torch.zeros(ctx.shape, dtype=torch.float32, device=g.device), None, None, None
        g = _lib.f32c(g, 'grad')
        d = torch.empty(max(B, 1), dtype=torch.float32, device=g.device)
        _lib.call('recnow_listwise_dense_bwd', _lib.ptr(g), _lib.ptr(row_rank), B, _lib.ptr(d), _lib.stream())
        return d[:B].reshape(ctx.shape), None, None, None


def to_listwise_sample(group_ids, labels, logits, do_mask_logits=True, value_of_masked_logit=-1E9, pos_neg_th=0.5):
    """Extract listwise samples from a batch (reference :89-148).

    A group is kept only if it has both a positive (label > pos_neg_th) and a negative (label - pos_neg_th < 0) row.
    Returns (dense_mask bool, dense_labels, dense_logits), each (num_valid_group, batch_size); rows follow the first
    occurrence of the group in the batch, columns are original row positions, labels are row-normalised, padded logits
    are `value_of_masked_logit` (0 when do_mask_logits=False).  Host-synchronises once (output size).
    """
    st = _list_stats(group_ids, labels, logits, do_mask_logits, value_of_masked_logit, pos_neg_th)
    B, dev, seg = st.B, st.seg.device, st.seg
    gv = int(st.n_valid.item())
    mask = torch.zeros((gv, B), dtype=torch.uint8, device=dev)
    dlabels = torch.zeros((gv, B), dtype=torch.float32, device=dev)
    dlogits = torch.full((gv, B), st.pad_logit, dtype=torch.float32, device=dev)
    # the fused kernel's by-product gives the row -> valid-rank map needed by backward
    loss, dbase, row_rank, group_loss = _row_rank(st)
    _lib.call('recnow_listwise_loss_fwdbwd', _lib.ptr(st.labels), _lib.ptr(st.logits), _lib.ptr(seg.order), _lib.ptr(seg.seg_id),
              _lib.ptr(seg.seg_first), _lib.ptr(st.seg_valid), _lib.ptr(st.seg_lse), _lib.ptr(st.seg_ysum), _lib.ptr(st.seg_psum),
              _lib.ptr(st.seg_pdot), _lib.ptr(st.valid_rank), _lib.ptr(st.n_valid), None, B, _lib.ptr(loss), _lib.ptr(dbase),
              _lib.ptr(row_rank), _lib.ptr(group_loss), _lib.stream())
    if gv > 0:
        _lib.call('recnow_listwise_dense', _lib.ptr(st.labels), _lib.ptr(st.logits), _lib.ptr(seg.order), _lib.ptr(seg.seg_id),
                  _lib.ptr(st.seg_ysum), _lib.ptr(st.valid_rank), B, _lib.ptr(mask), _lib.ptr(dlabels), _lib.ptr(dlogits),
                  _lib.stream())
    if isinstance(logits, torch.Tensor) and logits.requires_grad:
        dlogits = _DenseLogits.apply(logits, dlogits, row_rank, B)
    return mask.to(torch.bool), dlabels, dlogits


class _SoftmaxCERows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, labels, logits):
        labels = _lib.f32c(labels, 'labels_for_softmax')
        logits = _lib.f32c(logits, 'logits_for_softmax')
        if labels.shape != logits.shape or logits.dim() != 2:
            raise ValueError('labels and logits must be 2-D with the same shape')
        G, N = logits.shape
        dev = logits.device
        row_loss, row_lse, row_psum = (torch.empty(max(G, 1), dtype=torch.float32, device=dev) for _ in range(3))
        if G > 0:
            _lib.call('recnow_softmax_ce_rows_fwd', _lib.ptr(labels), _lib.ptr(logits), G, N, _lib.ptr(row_loss), _lib.ptr(row_lse),
                      _lib.ptr(row_psum), _lib.stream())
        ctx.save_for_backward(labels, logits, row_lse, row_psum)
        return row_loss[:G]

    @staticmethod
    def backward(ctx, g):
        labels, logits, row_lse, row_psum = ctx.saved_tensors
        G, N = logits.shape
        d = torch.empty_like(logits)
        if G > 0:
            g = _lib.f32c(g, 'grad')
            _lib.call('recnow_softmax_ce_rows_bwd', _lib.ptr(labels), _lib.ptr(logits), _lib.ptr(row_lse), _lib.ptr(row_psum),
                      _lib.ptr(g), G, N, _lib.ptr(d), _lib.stream())
        return None, d


def listwise_loss_via_softmax_cross_entropy_with_logits(labels_for_softmax,
                                                        logits_for_softmax,
                                                        weights=None,
                                                        do_reduce=True):
    """Softmax cross-entropy listwise loss on dense (G,B) samples (reference :151-173): labels are constants
    (stop_gradient), optional per-list weights, mean over lists with NaN -> 0 when do_reduce."""
    _lib.require_gpu(logits_for_softmax, 'logits_for_softmax')
    listwise_loss = _SoftmaxCERows.apply(labels_for_softmax.detach(), logits_for_softmax)
    if weights is not None:
        listwise_loss = listwise_loss * weights
    if do_reduce:
        if listwise_loss.numel() == 0:
            listwise_loss = listwise_loss.sum() + float('nan')       # reduce_mean of an empty tensor is NaN
        else:
            listwise_loss = listwise_loss.mean()
        listwise_loss = nan_to_zero(listwise_loss)
    return listwise_loss


class _ListwiseFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, group_ids, labels, weights, do_reduce, do_mask_logits, value_of_masked_logit, pos_neg_th):
        st = _list_stats(group_ids, labels, logits, do_mask_logits, value_of_masked_logit, pos_neg_th)
        seg, B = st.seg, st.B
        loss, dbase, row_rank, group_loss = _row_rank(st)
        w = None
        if weights is not None:
            w = _lib.f32c(weights, 'weights').reshape(-1)
            n_w = w.numel()
            if n_w < B:
                # the kernel reads weights[rank of a valid list]; the number of valid lists is data-dependent (<= B), so the
                # buffer handed over always holds B entries -- a too-short `weights` is reported below (do_reduce=False) or
                # weighs the lists beyond its end with 0, never an out-of-bounds read
                w = torch.cat([w, w.new_zeros(B - n_w)])
        _lib.call('recnow_listwise_loss_fwdbwd', _lib.ptr(st.labels), _lib.ptr(st.logits), _lib.ptr(seg.order), _lib.ptr(seg.seg_id),
                  _lib.ptr(seg.seg_first), _lib.ptr(st.seg_valid), _lib.ptr(st.seg_lse), _lib.ptr(st.seg_ysum), _lib.ptr(st.seg_psum),
                  _lib.ptr(st.seg_pdot), _lib.ptr(st.valid_rank), _lib.ptr(st.n_valid), _lib.ptr(w), B, _lib.ptr(loss), _lib.ptr(dbase),
                  _lib.ptr(row_rank), _lib.ptr(group_loss), _lib.stream())
        n_valid = st.n_valid.to(torch.float32).reshape(())
        ctx.shape = logits.shape
        ctx.do_reduce = do_reduce
        if do_reduce:
            ctx.save_for_backward(dbase[:B], n_valid, loss)
            ctx.mark_non_differentiable(n_valid)
            return loss, n_valid
        gv = int(st.n_valid.item())
        if weights is not None and weights.numel() != gv:
            raise ValueError('weights must have one entry per valid list: %d lists, %d weights' % (gv, weights.numel()))
        ctx.save_for_backward(dbase[:B], row_rank[:B])
        ctx.mark_non_differentiable(n_valid)
        return group_loss[:gv].clone(), n_valid

    @staticmethod
    def backward(ctx, g, _gn):
        if ctx.do_reduce:
            dbase, n_valid, loss = ctx.saved_tensors
            scale = torch.where(n_valid > 0, g / torch.clamp(n_valid, min=1.0), torch.zeros_like(g))
            d = dbase * scale
        else:
            dbase, row_rank = ctx.saved_tensors
            idx = torch.clamp(row_rank, min=0).long()
            gg = g[idx] if g.numel() > 0 else torch.zeros_like(dbase)
            d = torch.where(row_rank >= 0, dbase * gg, torch.zeros_like(dbase))
        return d.reshape(ctx.shape), None, None, None, None, None, None, None


class _ListwiseOneCall(torch.autograd.Function):
    """do_reduce=True through ONE library call (recnow_listwise_loss): grouping, list statistics, mean loss and the gradient already
    divided by the number of valid lists, on two allocations; the backward pass is one multiply."""

    @staticmethod
    def forward(ctx, logits, gkey, gdt, labels, weights, pad_logit, pos_neg_th):
        B = gkey.numel()
        lab = _lib.f32c(labels, 'labels').reshape(-1)
        lg = _lib.f32c(logits, 'logits').reshape(-1)
        if lab.numel() != B or lg.numel() != B:
            raise ValueError('group_ids, labels and logits must have the same number of elements')
        w = None
        if weights is not None:
            w = _lib.f32c(weights, 'weights').reshape(-1)
            if w.numel() < B:          # the kernel reads weights[rank of a valid list] (<= B lists): never past the buffer's end
                w = torch.cat([w, w.new_zeros(B - w.numel())])
        dev = gkey.device
        nws = _lib.load().recnow_listwise_loss_workspace_bytes(B, gdt)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        out = torch.empty(B + 2, dtype=torch.float32, device=dev)       # [d loss / d logits (B) | loss | number of valid lists]
        _lib.call('recnow_listwise_loss', _lib.ptr(gkey), gdt, _lib.ptr(lab), _lib.ptr(lg), _lib.ptr(w), B, float(pos_neg_th), float(pad_logit),
                  _lib._P(out.data_ptr() + 4 * B), _lib.ptr(out), _lib.ptr(ws), nws, _lib.stream())
        ctx.save_for_backward(out)
        ctx.B, ctx.shape = B, logits.shape
        n_valid = out[B + 1].reshape(())
        ctx.mark_non_differentiable(n_valid)
        return out[B].reshape(()), n_valid

    @staticmethod
    def backward(ctx, g, _gn):
        (out,) = ctx.saved_tensors
        return (out[:ctx.B] * g).reshape(ctx.shape), None, None, None, None, None, None


_LW_ONE_CALL = __import__('os').environ.get('RECNOW_LISTWISE_ONE_CALL', '1') != '0'      # A/B switch: '0' keeps the piecewise host route


def listwise_loss_from_batch(group_ids, labels, logits, weights=None, do_reduce=True, do_mask_logits=True,
                             value_of_masked_logit=-1E9, pos_neg_th=0.5, return_num_list=False):
    """Fused equivalent of
        mask, y, s = to_listwise_sample(group_ids, labels, logits, do_mask_logits, value_of_masked_logit, pos_neg_th)
        loss = listwise_loss_via_softmax_cross_entropy_with_logits(y, s, weights, do_reduce)
    without the (G,B) matrices and (for do_reduce=True) without a host sync.  `weights`: (num_valid_group,) in
    first-occurrence order of the valid groups.  Returns loss [, number of valid lists as a float tensor]."""
    if do_reduce and _LW_ONE_CALL and isinstance(group_ids, torch.Tensor) and group_ids.is_cuda and group_ids.numel() > 0:
        from ._segments import _as_key_tensor
        gkey, gdt = _as_key_tensor(group_ids)
        loss, n_valid = _ListwiseOneCall.apply(logits, gkey, gdt, labels, weights, float(value_of_masked_logit) if do_mask_logits else 0.0,
                                               pos_neg_th)
        return (loss, n_valid) if return_num_list else loss
    loss, n_valid = _ListwiseFused.apply(logits, group_ids, labels, weights, do_reduce, do_mask_logits, value_of_masked_logit,
                                         pos_neg_th)
    return (loss, n_valid) if return_num_list else loss
