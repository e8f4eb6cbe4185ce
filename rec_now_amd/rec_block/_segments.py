"""Group rows of a mini-batch by id on the GPU: canonical keys -> stable radix sort -> segments.

Host-side counterpart of `recnow_group_keys` / `recnow_group_segments` (include/recnow.h).  It replaces the dense
(B,B) mask of /root/reference/rec_now/rec_block/pairwise_loss_from_batch.py:16-74 and `tf.unique_with_counts` of
/root/reference/rec_now/rec_block/listwise_loss_from_batch.py:109.
"""
import torch

from .. import _lib

_KEY_F32, _KEY_F64, _KEY_I32, _KEY_I64 = 0, 1, 2, 3


def _as_key_tensor(g):
    if not isinstance(g, torch.Tensor):
        g = torch.as_tensor(g, device='cuda')
    _lib.require_gpu(g, 'group ids')
    g = g.reshape(-1)
    if g.dtype == torch.float32:
        return g.contiguous(), _KEY_F32
    if g.dtype == torch.float64:
        return g.contiguous(), _KEY_F64
    if g.dtype in (torch.float16, torch.bfloat16):
        return g.to(torch.float32).contiguous(), _KEY_F32
    if g.dtype == torch.int32:
        return g.contiguous(), _KEY_I32
    if g.dtype == torch.int64:
        return g.contiguous(), _KEY_I64
    if g.dtype in (torch.int8, torch.uint8, torch.int16, torch.bool):
        return g.to(torch.int32).contiguous(), _KEY_I32
    raise TypeError('unsupported group id dtype %s' % g.dtype)


class Segments(object):
    """Sorted-segment view of the batch.  All arrays are int32 CUDA tensors sized by B (no host sync)."""

    __slots__ = ('B', 'device', 'order', 'seg_id', 'seg_first', 'super_id', 'n_seg')

    def num_segments(self):
        """Host-synchronising read of the number of groups.  Raises if the single-launch grouping kernel reported a grid-barrier
        time-out (n_seg = -1; it leaves the identity grouping behind, so nothing downstream walks out of bounds)."""
        n = int(self.n_seg[0].item())
        if n < 0:
            raise RuntimeError('rec_now_amd: the cooperative grouping kernel timed out at a grid barrier (workgroups not co-resident); '
                               'set RECNOW_GROUP_COOP=0 to use the multi-launch route')
        return n


def build_segments(groups):
    """groups: tensor or list of tensors, each with B elements ((B,), (B,1) or (1,B)).  A list means "same group in
    EVERY tensor" (logical AND of the masks, pairwise_loss_from_batch.py:65-73); groups[0] is the main group used for
    the occurrence weights (:285)."""
    if not (isinstance(groups, (list, tuple)) and len(groups) > 0 and all(isinstance(g, torch.Tensor) for g in groups)):
        groups = [groups]          # one tensor, or one array-like of ids (e.g. a python list of numbers)
    if len(groups) == 0:
        raise ValueError('groups must not be empty')
    keyed = [_as_key_tensor(g) for g in groups]
    B = keyed[0][0].numel()
    dev = keyed[0][0].device
    for t, _ in keyed:
        if t.numel() != B:
            raise ValueError('all group tensors must have the same number of elements')
    lib = _lib.load()
    nws = [lib.recnow_key_words(dt) for _, dt in keyed]
    n_words = sum(nws)
    if n_words > 8:
        raise ValueError('at most 8 key words (e.g. 8 float32/int32 or 4 int64 group tensors) are supported')
    solo = torch.zeros(max(B, 1), dtype=torch.uint8, device=dev)
    st = _lib.stream()
    if len(keyed) == 1 and keyed[0][1] == _KEY_I32 and B > 0:
        # ONE int32 id tensor (the pooled lookup's 6.5 M sort keys): its bits ARE the key words -- recnow_group_keys would copy them (18 us there)
        words = keyed[0][0].reshape(1, -1)
    else:
        words = torch.empty((n_words, max(B, 1)), dtype=torch.int32, device=dev)
        off = 0
        for (t, dt), nw in zip(keyed, nws):
            _lib.call('recnow_group_keys', _lib.ptr(t), dt, B, _lib.ptr(words[off:]), _lib.ptr(solo), st)
            off += nw
    seg = Segments()
    seg.B, seg.device = B, dev
    seg.order = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    seg.seg_id = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    seg.seg_first = torch.empty(B + 1, dtype=torch.int32, device=dev)
    seg.super_id = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    seg.n_seg = torch.empty(2, dtype=torch.int32, device=dev)
    nbytes = lib.recnow_group_segments_workspace_bytes(B, n_words)
    ws = _lib.workspace(nbytes, dev)
    _lib.call('recnow_group_segments', _lib.ptr(words), _lib.ptr(solo), B, n_words, nws[0], _lib.ptr(seg.order),
              _lib.ptr(seg.seg_id), _lib.ptr(seg.seg_first), _lib.ptr(seg.super_id), _lib.ptr(seg.n_seg), _lib.ptr(ws),
              ws.numel(), st)
    return seg
