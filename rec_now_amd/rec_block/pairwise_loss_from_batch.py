"""In-batch pairwise ranking loss -- drop-in for rec_now/rec_block/pairwise_loss_from_batch.py.

Same public names, argument order, defaults and return structure as the reference
(/root/reference/rec_now/rec_block/pairwise_loss_from_batch.py), with torch.Tensor in place of tf.Tensor.  Underneath,
the reference's dense O(B^2) masks are replaced by sort-by-group + segmented pair enumeration in hand-written HIP
kernels (csrc/scan_sort.hip, csrc/pairwise.hip); pair order, counts and weights are identical to the reference's.

Two execution paths:
  * fused   -- `pairloss_func is bpr_loss_func` and no `label_pair_to_weight_func`: loss and d(loss)/d(outputs) come
               out of one kernel, pairs are never materialised;
  * general -- any other callable: pairs are materialised (bit-exact reference order), labels/outputs are gathered to
               (P,) vectors and the user's callables run on those.  Exact for element-wise callables (what the
               reference's own test uses, tests/rec_block/test_pairwise_loss_from_batch.py:51-53); callables that
               inspect the (B,B) *shape* are not supported.
"""
import torch

from .. import _lib
from ._segments import _as_key_tensor, build_segments

SMALL_POSIVITE_FLOAT = 1.0E-10   # reference :13 (spelling kept)

import os as _os

_ONE_CALL = _os.environ.get('RECNOW_PAIR_ONE_CALL', '1') != '0'      # A/B switch: '0' keeps the piecewise host route
_FLAG_LABEL_GT, _FLAG_WRONG_ORDER = 1, 2
_FLAG_MEMBERS_PACKED = 256          # RECNOW_PAIR_MEMBERS_PACKED: the workspace of _count is handed straight to the loss kernel


def _generate_pair_mask(sample_group_idx_var, only_upper_band=False):
    """(B,B) bool mask of same-group, off-diagonal pairs (reference :16-40)."""
    return generate_pair_mask(sample_group_idx_var, only_upper_band)


def generate_pair_mask(group_tensor_or_list, only_upper_band=False):
    """Dense (B,B) bool mask, True where rows i != j share a group in every tensor of the list (reference :43-74).
    `only_upper_band=True` keeps only the first super-diagonal, exactly like tf.linalg.band_part(m, 0, 1) (:38-39).
    Provided for API parity; `pairwise_loss` itself never builds this matrix."""
    groups = list(group_tensor_or_list) if isinstance(group_tensor_or_list, (list, tuple)) else [group_tensor_or_list]
    seg = build_segments(groups)
    B = seg.B
    out = torch.zeros((B, B), dtype=torch.uint8, device=seg.device)
    _lib.call('recnow_pair_mask_dense', _lib.ptr(seg.order), _lib.ptr(seg.seg_id), _lib.ptr(seg.seg_first), B,
              1 if only_upper_band else 0, _lib.ptr(out), _lib.stream())
    return out.to(torch.bool)


def vec_to_matrix_pair(vec):
    """(B,1)/(1,B) vector -> (M, M^T) with M[i,j] = v_i (reference :77-93).  Returned as broadcast views."""
    vec = vec.reshape(-1, 1)
    mat = vec.expand(vec.shape[0], vec.shape[0])
    return mat, mat.t()


class _BprVec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg, weights, factor, reduce_mean):
        shape = pos.shape
        p = _lib.f32c(pos, 'outputs_pos').reshape(-1)
        n = _lib.f32c(neg, 'outputs_neg').reshape(-1)
        if p.numel() != n.numel():
            raise ValueError('outputs_pos and outputs_neg must have the same number of elements')
        w = None
        if weights is not None:
            w = _lib.f32c(weights, 'weights').reshape(-1)
            if w.numel() != p.numel():
                w = w.expand_as(p).contiguous()
        P = p.numel()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        dpos = torch.empty(max(P, 1), dtype=torch.float32, device=p.device)
        ws = _lib.workspace(1024 * 8 + 256, p.device)
        _lib.call('recnow_bpr_loss_fwdbwd', _lib.ptr(p), _lib.ptr(n), _lib.ptr(w), P, float(factor),
                  1 if reduce_mean else 0, _lib.ptr(loss), _lib.ptr(dpos), _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.save_for_backward(dpos[:P])
        ctx.shape = shape
        ctx.neg_shape = neg.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpos,) = ctx.saved_tensors
        gp = dpos * g
        return gp.reshape(ctx.shape), (-gp).reshape(ctx.neg_shape), None, None, None


def bpr_loss_func(outputs_pos, outputs_neg, weights=None, factor=1.0, reduce_mean=True):
    """BPR / logistic pair loss on explicit vectors (reference :96-127):
    sum(w * softplus(-factor*(pos-neg))) / (P + 1e-10)   (raw sum when reduce_mean=False).  weights are constants."""
    return _BprVec.apply(outputs_pos, outputs_neg, weights, factor, reduce_mean)


def occurance_power_weight(group_id, power=0.0):
    """weights[i] = (number of elements sharing group_id[i]) ** power (reference :130-151)."""
    seg = build_segments(group_id)
    w = torch.empty(max(seg.B, 1), dtype=torch.float32, device=seg.device)
    _lib.call('recnow_occurance_power_weight', _lib.ptr(seg.order), _lib.ptr(seg.seg_id), _lib.ptr(seg.seg_first), seg.B,
              float(power), _lib.ptr(w), _lib.stream())
    return w[:seg.B]


def _flat_f32(t, B, what):
    t = _lib.f32c(t, what).reshape(-1)
    if t.numel() != B:
        raise ValueError('%s must have %d elements, got %d' % (what, B, t.numel()))
    return t


def _flat_mask(mask, B):
    if mask is None:
        return None
    _lib.require_gpu(mask, 'mask')
    m = (mask.reshape(-1) != 0).to(torch.uint8).contiguous()
    if m.numel() != B:
        raise ValueError('mask must have %d elements' % B)
    return m


def _count(scores, labels, mask, seg, flags):
    B, dev = seg.B, seg.device
    cnt_row = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    cnt_super = torch.empty(max(B, 1), dtype=torch.int64, device=dev)
    n_pair = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _lib.workspace(_lib.load().recnow_pairwise_workspace_bytes(B), dev)
    _lib.call('recnow_pair_count', _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(mask), _lib.ptr(seg.order),
              _lib.ptr(seg.seg_id), _lib.ptr(seg.seg_first), _lib.ptr(seg.super_id), B, flags, _lib.ptr(cnt_row),
              _lib.ptr(cnt_super), _lib.ptr(n_pair), _lib.ptr(ws), ws.numel(), _lib.stream())
    return cnt_row, cnt_super, n_pair, ws


def pair_indices(outputs, labels, groups, only_use_wrong_order_pair=False, mask=None, use_label_cond=True):
    """(pos_idx, neg_idx) int32 tensors of the surviving pairs in the reference's order: ascending positive row i,
    then ascending negative row j (= tf.boolean_mask over the row-major flattened (B,B) mask, reference :217,272-273).
    Host-synchronises once to size the output."""
    seg = build_segments(groups)
    B = seg.B
    scores = _flat_f32(outputs, B, 'outputs').detach()
    labs = _flat_f32(labels, B, 'labels').detach()
    m = _flat_mask(mask, B)
    flags = (_FLAG_LABEL_GT if use_label_cond else 0) | (_FLAG_WRONG_ORDER if only_use_wrong_order_pair else 0)
    cnt_row, _, n_pair, ws = _count(scores, labs, m, seg, flags)
    offsets = torch.empty(B + 1, dtype=torch.int64, device=seg.device)
    _lib.call('recnow_pair_offsets', _lib.ptr(cnt_row), B, _lib.ptr(offsets), _lib.ptr(ws), ws.numel(), _lib.stream())
    P = int(n_pair.item())
    pos = torch.empty(max(P, 1), dtype=torch.int32, device=seg.device)
    neg = torch.empty(max(P, 1), dtype=torch.int32, device=seg.device)
    _lib.call('recnow_pair_emit', _lib.ptr(scores), _lib.ptr(labs), _lib.ptr(m), _lib.ptr(seg.order), _lib.ptr(seg.seg_id),
              _lib.ptr(seg.seg_first), B, flags, _lib.ptr(offsets), _lib.ptr(pos), _lib.ptr(neg), P, _lib.ptr(ws),
              ws.numel(), _lib.stream())
    return pos[:P], neg[:P]


def _onepass(ctx, outputs, scores, labs, m, order, seg_id, seg_first, B, dev, flags, factor, reduce_mean, ws, n_pair):
    """Loss without occurrence weights (click_occurance_power == 0): counts and BPR terms from ONE walk per row
    (recnow_pair_bpr_onepass); the gradient stays unnormalised until the incoming gradient is multiplied in (backward)."""
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dscores = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    _lib.call('recnow_pair_bpr_onepass', _lib.ptr(scores), _lib.ptr(labs), _lib.ptr(m), _lib.ptr(order), _lib.ptr(seg_id),
              _lib.ptr(seg_first), B, flags, float(factor), 1 if reduce_mean else 0, _lib.ptr(loss), _lib.ptr(dscores), _lib.ptr(n_pair),
              _lib.ptr(ws), ws.numel(), _lib.stream())
    ctx.save_for_backward(dscores[:B], n_pair)
    ctx.shape = outputs.shape
    ctx.onepass = True
    ctx.reduce_mean = bool(reduce_mean)
    return loss, _n_pair_out(ctx, n_pair)


def _n_pair_out(ctx, n_pair):
    """The pair count as the float32 0-dim tensor the reference returns -- only when the caller asked for it (one conversion kernel
    and its host time less on the default `pairwise_loss(...)` path)."""
    if not ctx.want_np:
        return None
    n_pair_f = n_pair.to(torch.float32).reshape(())
    ctx.mark_non_differentiable(n_pair_f)
    return n_pair_f


def _onepass_backward(ctx, g):
    dscores, n_pair = ctx.saved_tensors
    gd = g.detach().to(torch.float32).reshape(1).contiguous()
    out = torch.empty_like(dscores)
    _lib.call('recnow_pair_scale_grad', _lib.ptr(dscores), _lib.ptr(gd), _lib.ptr(n_pair) if ctx.reduce_mean else None, 1.0e-10,
              dscores.numel(), _lib.ptr(out), _lib.stream())
    return out.reshape(ctx.shape)


class _PairBprFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, labels, mask, seg, flags, factor, power, reduce_mean, want_np=True):
        B = seg.B
        ctx.want_np = bool(want_np)
        scores = _flat_f32(outputs, B, 'outputs')
        labs = _flat_f32(labels, B, 'labels')
        m = _flat_mask(mask, B)
        ctx.onepass = False
        if float(power) == 0.0:
            n_pair = torch.empty(1, dtype=torch.int64, device=seg.device)
            ws = _lib.workspace(_lib.load().recnow_pairwise_workspace_bytes(B), seg.device)
            return _onepass(ctx, outputs, scores, labs, m, seg.order, seg.seg_id, seg.seg_first, B, seg.device, flags, factor,
                            reduce_mean, ws, n_pair)
        cnt_row, cnt_super, n_pair, ws = _count(scores, labs, m, seg, flags)
        loss = torch.empty((), dtype=torch.float32, device=seg.device)
        dscores = torch.empty(max(B, 1), dtype=torch.float32, device=seg.device)
        _lib.call('recnow_pair_bpr_fwdbwd', _lib.ptr(scores), _lib.ptr(labs), _lib.ptr(m), _lib.ptr(seg.order),
                  _lib.ptr(seg.seg_id), _lib.ptr(seg.seg_first), _lib.ptr(seg.super_id), _lib.ptr(cnt_super),
                  _lib.ptr(n_pair), B, flags | _FLAG_MEMBERS_PACKED, float(factor), float(power), 1 if reduce_mean else 0, _lib.ptr(loss),
                  _lib.ptr(dscores), _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.save_for_backward(dscores[:B])
        ctx.shape = outputs.shape
        return loss, _n_pair_out(ctx, n_pair)

    @staticmethod
    def backward(ctx, g, _g_np):
        if ctx.onepass:
            return _onepass_backward(ctx, g), None, None, None, None, None, None, None, None
        (dscores,) = ctx.saved_tensors
        return (dscores * g).reshape(ctx.shape), None, None, None, None, None, None, None, None


class _PairBprSmall(torch.autograd.Function):
    """B <= 8192 rows, one float32 / int32 group tensor: keys, grouping and member packing in ONE launch
    (recnow_group_pack_small), then the counting and loss kernels of the general route on the packed members."""

    @staticmethod
    def forward(ctx, outputs, labels, mask, gkey, gdt, flags, factor, power, reduce_mean, want_np=True):
        B = gkey.numel()
        ctx.want_np = bool(want_np)
        scores = _flat_f32(outputs, B, 'outputs')
        labs = _flat_f32(labels, B, 'labels')
        m = _flat_mask(mask, B)
        dev = gkey.device
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)      # noqa: E731
        order, seg_id, seg_first, super_id, n_seg = i32(max(B, 1)), i32(max(B, 1)), i32(B + 1), i32(max(B, 1)), i32(2)
        cnt_row = i32(max(B, 1))
        cnt_super = torch.empty(max(B, 1), dtype=torch.int64, device=dev)
        n_pair = torch.empty(1, dtype=torch.int64, device=dev)
        ws = _lib.workspace(_lib.load().recnow_pairwise_workspace_bytes(B), dev)
        st = _lib.stream()
        _lib.call('recnow_group_pack_small', _lib.ptr(gkey), gdt, _lib.ptr(labs), _lib.ptr(scores), _lib.ptr(m), B, _lib.ptr(order),
                  _lib.ptr(seg_id), _lib.ptr(seg_first), _lib.ptr(super_id), _lib.ptr(n_seg), _lib.ptr(cnt_super), _lib.ptr(n_pair),
                  _lib.ptr(ws), ws.numel(), st)
        ctx.onepass = False
        if float(power) == 0.0:
            return _onepass(ctx, outputs, scores, labs, m, order, seg_id, seg_first, B, dev, flags | _FLAG_MEMBERS_PACKED, factor,
                            reduce_mean, ws, n_pair)
        _lib.call('recnow_pair_count', _lib.ptr(scores), _lib.ptr(labs), _lib.ptr(m), _lib.ptr(order), _lib.ptr(seg_id),
                  _lib.ptr(seg_first), _lib.ptr(super_id), B, flags | _FLAG_MEMBERS_PACKED, _lib.ptr(cnt_row), _lib.ptr(cnt_super),
                  _lib.ptr(n_pair), _lib.ptr(ws), ws.numel(), st)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dscores = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
        _lib.call('recnow_pair_bpr_fwdbwd', _lib.ptr(scores), _lib.ptr(labs), _lib.ptr(m), _lib.ptr(order), _lib.ptr(seg_id),
                  _lib.ptr(seg_first), _lib.ptr(super_id), _lib.ptr(cnt_super), _lib.ptr(n_pair), B, flags | _FLAG_MEMBERS_PACKED,
                  float(factor), float(power), 1 if reduce_mean else 0, _lib.ptr(loss), _lib.ptr(dscores), _lib.ptr(ws), ws.numel(), st)
        ctx.save_for_backward(dscores[:B])
        ctx.shape = outputs.shape
        return loss, _n_pair_out(ctx, n_pair)

    @staticmethod
    def backward(ctx, g, _g_np):
        if ctx.onepass:
            return _onepass_backward(ctx, g), None, None, None, None, None, None, None, None, None
        (dscores,) = ctx.saved_tensors
        return (dscores * g).reshape(ctx.shape), None, None, None, None, None, None, None, None, None


class _PairLossOneCall(torch.autograd.Function):
    """The default pairwise_loss (BPR, no occurrence weights, one group tensor) through ONE library call (recnow_pairwise_loss):
    grouping, loss and the gradient in a single ctypes call on two allocations (workspace; [d loss / d scores | loss | P]), the
    backward pass is one multiply.  Same kernels as the piecewise route; the host side of a B = 8192 loss drops from ~0.18 to ~0.1 ms."""

    @staticmethod
    def forward(ctx, outputs, labels, mask, gkey, gdt, flags, factor, reduce_mean, want_np):
        B = gkey.numel()
        scores = _flat_f32(outputs, B, 'outputs')
        labs = _flat_f32(labels, B, 'labels')
        m = _flat_mask(mask, B)
        dev = gkey.device
        lib = _lib.load()
        nws = lib.recnow_pairwise_loss_workspace_bytes(B, gdt)
        ws = torch.empty(nws + 8, dtype=torch.uint8, device=dev)          # + the int64 pair count at its (256-byte aligned) end
        out = torch.empty(B + 2, dtype=torch.float32, device=dev)         # [dscores (B) | loss | (float) P]
        n_pair_ptr = ws.data_ptr() + nws
        _lib.call('recnow_pairwise_loss', _lib.ptr(gkey), gdt, _lib.ptr(labs), _lib.ptr(scores), _lib.ptr(m), B, flags, float(factor),
                  1 if reduce_mean else 0, _lib._P(out.data_ptr() + 4 * B), _lib._P(n_pair_ptr), _lib._P(out.data_ptr() + 4 * B),
                  _lib.ptr(out), _lib.ptr(ws), nws, _lib.stream())
        ctx.save_for_backward(out)
        ctx.B = B
        ctx.shape = outputs.shape
        loss = out[B].reshape(())
        if not want_np:
            return loss, None
        n_pair_f = out[B + 1].reshape(())
        ctx.mark_non_differentiable(n_pair_f)
        return loss, n_pair_f

    @staticmethod
    def backward(ctx, g, _g_np):
        (out,) = ctx.saved_tensors
        return (out[:ctx.B] * g).reshape(ctx.shape), None, None, None, None, None, None, None, None


def _one_call_route(groups):
    """(key tensor, dtype code) when the one-call loss applies: ONE group tensor (any supported id dtype)."""
    if isinstance(groups, (list, tuple)):
        if len(groups) != 1:
            return None
        groups = groups[0]
    if not isinstance(groups, torch.Tensor) or not groups.is_cuda:
        return None
    try:
        return _as_key_tensor(groups)
    except TypeError:
        return None


def _small_route(groups):
    """(key tensor, dtype code) when the single-launch route applies: one group tensor of B <= 8192 float / int32-able ids."""
    if isinstance(groups, (list, tuple)):
        if len(groups) != 1:
            return None
        groups = groups[0]
    if not isinstance(groups, torch.Tensor) or not groups.is_cuda or groups.dtype in (torch.float64, torch.int64):
        return None
    try:
        key, dt = _as_key_tensor(groups)
    except TypeError:
        return None
    if not _lib.load().recnow_pairwise_small_supported(key.numel(), dt):
        return None
    return key, dt


def group_rows(groups):
    """The score-independent half of the loss: canonical keys -> radix sort -> segments of `groups` (tensor or list of
    tensors, as `pairwise_loss` takes them).  The result can be handed to `pairwise_loss_fused(..., segments=...)`, so a
    training step can build it on a side stream while the model's forward pass is still producing the scores."""
    return build_segments(groups)


def pairwise_loss_fused(outputs, labels, groups, only_use_wrong_order_pair=False, click_occurance_power=0.0, mask=None,
                        factor=1.0, reduce_mean=True, segments=None, return_num_pair=True):
    """Fused BPR pairwise loss; returns (loss, n_pair) as 0-dim tensors, no host sync.  `pairwise_loss` routes here
    whenever the defaults make it possible; exposed because it also accepts `factor` / `reduce_mean` and a precomputed
    `segments=group_rows(groups)` (then `groups` is not looked at again).  return_num_pair=False: the second value is None (the
    float32 conversion of the pair count is a kernel of its own)."""
    flags = _FLAG_LABEL_GT | (_FLAG_WRONG_ORDER if only_use_wrong_order_pair else 0)
    if segments is None and float(click_occurance_power) == 0.0 and _ONE_CALL:
        one = _one_call_route(groups)
        if one is not None and one[0].numel() > 0:
            return _PairLossOneCall.apply(outputs, labels, mask, one[0], one[1], flags, factor, reduce_mean, return_num_pair)
    if segments is None:
        small = _small_route(groups)
        if small is not None:
            return _PairBprSmall.apply(outputs, labels, mask, small[0], small[1], flags, factor, click_occurance_power, reduce_mean,
                                       return_num_pair)
    seg = segments if segments is not None else build_segments(groups)
    return _PairBprFused.apply(outputs, labels, mask, seg, flags, factor, click_occurance_power, reduce_mean, return_num_pair)


def _merge_weights_by_mul(weights1, weights2):
    if weights1 is None:
        return weights2
    if weights2 is None:
        return weights1
    return weights1 * weights2


def pairwise_loss(outputs, labels, groups,
                  pairloss_func=bpr_loss_func,
                  only_use_wrong_order_pair=False,
                  return_num_pair=False,
                  click_occurance_power=0.0,
                  mask=None,
                  label_pair_to_weight_func=None,
                  **kwargs
                  ):
    """Pairwise loss over all in-batch pairs (i, j) of the same group with label_i > label_j (reference :228-279).

    Args (as the reference): outputs, labels: (B,)/(B,1) tensors; groups: tensor or list of tensors (AND of the
    conditions, groups[0] = main group for `click_occurance_power`); pairloss_func(outputs_pos, outputs_neg, weights);
    only_use_wrong_order_pair; return_num_pair; click_occurance_power; mask (bool, same shape as labels);
    label_pair_to_weight_func(label_pos, label_neg, **kwargs) -> weights, pairs with weight <= 0 are dropped.
    Returns: loss, or (loss, n_pair as float32 tensor) when return_num_pair.
    """
    if pairloss_func is bpr_loss_func and label_pair_to_weight_func is None:
        loss, n_pair = pairwise_loss_fused(outputs, labels, groups, only_use_wrong_order_pair, click_occurance_power, mask,
                                           return_num_pair=return_num_pair)
        return (loss, n_pair) if return_num_pair else loss

    # general path: materialise the pair list, then run the user's callables on (P,) vectors
    flat_out = _lib.require_gpu(outputs, 'outputs').reshape(-1)
    flat_lab = _lib.require_gpu(labels, 'labels').reshape(-1)
    pos, neg = pair_indices(outputs, labels, groups, only_use_wrong_order_pair, mask,
                            use_label_cond=label_pair_to_weight_func is None)
    weights = None
    if label_pair_to_weight_func is not None:
        w_all = label_pair_to_weight_func(flat_lab[pos.long()], flat_lab[neg.long()], **kwargs)   # reference :192
        keep = w_all > 0                                                                        # reference :193
        pos, neg, weights = pos[keep], neg[keep], w_all[keep]
    if click_occurance_power != 0.0:                                                            # reference :282-291
        group = groups[0] if isinstance(groups, (list, tuple)) else groups
        if pos.numel() > 0:
            occ = occurance_power_weight(group.reshape(-1)[pos.long()], power=click_occurance_power)
        else:
            occ = torch.empty(0, dtype=torch.float32, device=flat_out.device)
        weights = _merge_weights_by_mul(weights, occ)
    if weights is not None:
        weights = weights.detach()                                                              # reference :269-270
    outputs_pos = flat_out[pos.long()]
    outputs_neg = flat_out[neg.long()]
    loss = pairloss_func(outputs_pos, outputs_neg, weights)                                     # reference :274
    if return_num_pair:
        return loss, torch.tensor(float(pos.numel()), dtype=torch.float32, device=flat_out.device)
    return loss
