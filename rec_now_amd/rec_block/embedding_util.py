"""Pooled embedding lookup by slot -- drop-in for the pooled-lookup path of rec_now/rec_block/embedding_util.py
(/root/reference/rec_now/rec_block/embedding_util.py:138-195 `sparse_batch_segment_ids_of_targets` and :239-324
`embedding_using_sparse_batch_segment_ids`): the step that produces the (B, T, D) field embeddings the interaction layers
consume.  Host side of the HIP kernels in csrc/embed.hip; ids are sorted / uniqued with the same radix-sort machinery as
the in-batch losses (rec_block/_segments.py).
"""
import torch

from .. import _lib
from ._segments import build_segments

_KEY_I32, _KEY_I64 = 2, 3


def _slot_tensor(slots):
    if not isinstance(slots, torch.Tensor):
        slots = torch.as_tensor(slots, device='cuda')
    _lib.require_gpu(slots, 'slots')
    if slots.dtype == torch.int64:
        return slots.contiguous(), _KEY_I64, torch.int64
    if slots.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
        return slots.to(torch.int32).contiguous(), _KEY_I32, torch.int32
    raise TypeError('slots must be an integer tensor (string slots of the reference are hashed upstream); got %s' % slots.dtype)


def _slot_targets(slots, target_slots, ids, want_key, id_limit=0):
    """(B,C) slots -> seg (B,C) int32 target index or -1, and optionally the int64 sort key (id, or KEY_NOT_POOLED).
    id_limit = V (a table of V < 2^31 - 1 rows): entries that are not pooled and ids outside the table carry the key V instead, and a
    third value -- the same keys as int32 -- is returned for the sort (one key word: three digit passes for V = 2^20 instead of four)."""
    if not isinstance(target_slots, list):
        target_slots = list(target_slots)
    if len(set(target_slots)) != len(target_slots):
        raise ValueError('target_slots must not contain duplicates')       # the reference's StaticHashTable rejects them too
    slots, sdt, tdt = _slot_tensor(slots)
    if slots.dim() != 2:
        raise ValueError('slots must be a (B, C) matrix')
    dev = slots.device
    targets = _lib.const_array(target_slots, tdt, dev)
    seg = torch.empty(slots.shape, dtype=torch.int32, device=dev)
    key = torch.empty(slots.shape, dtype=torch.int64, device=dev) if want_key else None
    narrow = bool(want_key and 0 < id_limit < (1 << 31) - 1)
    key32 = torch.empty(slots.shape, dtype=torch.int32, device=dev) if narrow else None
    _lib.call('recnow_slot_targets', _lib.ptr(slots), sdt, _lib.ptr(targets), len(target_slots), _lib.ptr(ids) if want_key else None,
              slots.numel(), _lib.ptr(seg), _lib.ptr(key), int(id_limit) if narrow else 0, _lib.ptr(key32), _lib.stream())
    if id_limit:
        return seg, key, key32
    return seg, key


def sparse_batch_segment_ids_of_targets(slots, target_slots):
    """embedding_util.py:138-195.  Returns (mask (B,C) bool, sp_segment_ids (n,) int32 = row * T + target index of the
    masked-in entries in row-major order, num_rows, num_ids, num_segments).  The compaction has a data-dependent size, so
    this (API-parity) function synchronises; the pooled lookup below never materialises it."""
    if not isinstance(target_slots, list):
        target_slots = list(target_slots)
    seg, _ = _slot_targets(slots, target_slots, None, False)
    B, _C = seg.shape
    T = len(target_slots)
    mask = seg >= 0
    rows = torch.arange(B, dtype=torch.int32, device=seg.device).reshape(-1, 1) * T
    sp = (seg + rows)[mask]
    return mask, sp, B, T, B * T


KEY_NOT_POOLED = -(1 << 63)      # sort key of entries that are not pooled (include/recnow.h, recnow_slot_targets)


class EmbeddingTable(torch.nn.Module):
    """`embedding_func` backed by one dense (V, D) table (the reference docstring's own example, :254-256:
    `tf.nn.embedding_lookup(params, ids)`).  Passing an instance to embedding_using_sparse_batch_segment_ids selects the
    fused path: the pooled rows are gathered straight from the table, no unique/gather round trip."""

    def __init__(self, params, sparse_grad=False):
        """sparse_grad: hand the table's gradient back as a torch.sparse_coo_tensor of the rows that were looked up (what the
        reference path yields: tf.IndexedSlices) instead of a dense, zero-filled (V, D) tensor -- for large tables the dense form is
        hundreds of MB of memset and optimizer traffic per step.  Costs one host sync per backward (the number of distinct ids is
        data-dependent); use with an optimizer that accepts sparse gradients (torch.optim.SparseAdam, SGD)."""
        super().__init__()
        self.weight = params if isinstance(params, torch.nn.Parameter) else torch.nn.Parameter(torch.as_tensor(params, dtype=torch.float32))
        self.sparse_grad = bool(sparse_grad)

    def forward(self, ids):
        _lib.require_gpu(ids, 'ids')
        return _lookup_rows(self.weight, ids.reshape(-1).to(torch.int64)).reshape(tuple(ids.shape) + (self.weight.shape[1],))


class _Sorted(object):
    """Entries sorted by id (lazy: only the backward of a trainable table, or the unique path, needs it)."""

    def __init__(self, key, key32=None):
        self.key = key
        self.key32 = key32          # the same keys as int32 (table path, V < 2^31 - 1): what the sort runs on
        self._seg = None

    def segments(self):
        if self._seg is None:
            self._seg = build_segments((self.key if self.key32 is None else self.key32).reshape(-1))
        return self._seg


class _PoolFunction(torch.autograd.Function):
    """out (B,T,D) = segment-sum/mean over the pooled entries of weights * table[rows].  `table` is either the full
    embedding table (rows = ids; its gradient is scattered into a dense (V,D) tensor) or the (U,D) embeddings of the
    unique ids (rows = inverse index; the gradient comes out dense in unique order)."""

    @staticmethod
    def forward(ctx, table, rows, seg, weights, T, mean, srt, dense_scatter):
        table = _lib.f32c(table, 'embedding table')
        B, C = seg.shape
        D = table.shape[1]
        dev = seg.device
        out = torch.empty((B, T, D), dtype=torch.float32, device=dev)       # every element is written by the kernel
        cnt = torch.empty((B, T), dtype=torch.float32, device=dev) if mean else None
        _lib.call('recnow_embed_pool_fwd', _lib.ptr(table), D, table.shape[0], _lib.ptr(rows), _lib.ptr(seg), _lib.ptr(weights), B, C, T,
                  1 if mean else 0, _lib.ptr(out), _lib.ptr(cnt), _lib.stream())
        need_dw = weights is not None and ctx.needs_input_grad[3]
        ctx.save_for_backward(seg, weights, cnt, *((table, rows) if need_dw else ()))
        ctx.meta = (T, D, bool(mean), srt, dense_scatter, table.shape[0], need_dw)
        return out

    @staticmethod
    def backward(ctx, dout):
        seg, weights, cnt, *extra = ctx.saved_tensors
        T, D, mean, srt, dense_scatter, V, need_dw = ctx.meta
        B, C = seg.shape
        N = B * C
        dev = seg.device
        dout = _lib.f32c(dout, 'grad')
        dweights = None
        if need_dw:             # TF autodiff of `embeddings * expand_dims(sp_weights, -1)` (reference :315-317)
            table, rows = extra
            dweights = torch.empty((B, C), dtype=torch.float32, device=dev)
            _lib.call('recnow_embed_pool_bwd_weights', _lib.ptr(table), D, V, _lib.ptr(rows), _lib.ptr(seg), _lib.ptr(cnt), _lib.ptr(dout),
                      B, C, T, 1 if mean else 0, _lib.ptr(dweights), _lib.stream())
        if not ctx.needs_input_grad[0]:
            return None, None, None, dweights, None, None, None, None
        s = srt.segments()
        drows = torch.empty((max(N, 1), D), dtype=torch.float32, device=dev)
        row_ids = torch.empty(max(N, 1), dtype=torch.int64, device=dev)
        ws = _lib.workspace(_lib.load().recnow_embed_rows_bwd_workspace_bytes(N, D), dev)
        _lib.call('recnow_embed_rows_bwd', _lib.ptr(srt.key), _lib.ptr(s.order), _lib.ptr(s.seg_id), _lib.ptr(s.seg_first),
                  _lib.ptr(s.n_seg), _lib.ptr(seg), _lib.ptr(weights), _lib.ptr(cnt), _lib.ptr(dout), N, C, T, D, 1 if mean else 0,
                  _lib.ptr(drows), _lib.ptr(row_ids), _lib.ptr(ws), ws.numel(), _lib.stream())
        if dense_scatter == 'sparse':
            n_used = s.num_segments()                    # distinct keys (incl. the one "not pooled" key, which sorts last)
            ids = row_ids[:n_used]
            keep = (ids >= 0) & (ids < V)                       # drops the not-pooled key and ids outside the table
            dtable = torch.sparse_coo_tensor(ids[keep].unsqueeze(0), drows[:n_used][keep], (V, D))
        elif dense_scatter:
            dtable = torch.zeros((V, D), dtype=torch.float32, device=dev)
            _lib.call('recnow_embed_scatter_rows', _lib.ptr(drows), _lib.ptr(row_ids), N, D, V, _lib.ptr(dtable), _lib.ptr(s.n_seg), _lib.stream())
        else:
            dtable = drows[:V]                      # unique path: segment s IS unique id s
        return dtable, None, None, dweights, None, None, None, None


def _lookup_rows(table, ids):
    """Plain row lookup table[ids] (EmbeddingTable called directly) = pooling with one entry per output row."""
    n = ids.numel()
    ids2 = ids.reshape(n, 1).contiguous()
    seg = torch.zeros((n, 1), dtype=torch.int32, device=ids.device)
    return _PoolFunction.apply(table, ids2, seg, None, 1, False, _Sorted(ids2), True).reshape(n, table.shape[1])


def embedding_using_sparse_batch_segment_ids(embedding_func, slots, target_slots, ids, weights=None, method='sum', use_unique=True):
    """Embed ids and pool them by slot: out[b][t] = sum (or mean) over the columns c of row b whose slot is
    target_slots[t] of weights[b][c] * embedding(ids[b][c]).

    Args:
        embedding_func: an EmbeddingTable (fused path), or any callable mapping a 1-D int64 id tensor to (n, D) embeddings.
        slots: (B, C) integer slots of the ids;  target_slots: list of T slots to pool;  ids: (B, C) ids (>= 0).
        weights: optional (B, C) per-id weights; differentiable (d out / d weights as TF autodiff gives it).
        method: 'sum' or 'mean';  use_unique: look each distinct id up once (:305-311) - only matters for a callable
            embedding_func, the fused table path never gathers an id it does not pool.
    Returns:
        pooled_embedding (B, T, D).
    """
    if method not in ('sum', 'mean'):
        raise ValueError("method must be 'sum' or 'mean'")
    if not isinstance(target_slots, list):
        target_slots = list(target_slots)
    if not isinstance(ids, torch.Tensor):
        ids = torch.as_tensor(ids, device='cuda')
    _lib.require_gpu(ids, 'ids')
    ids = ids.to(torch.int64).contiguous()
    if weights is not None:
        weights = _lib.f32c(weights, 'weights')
        if weights.shape != ids.shape:
            raise ValueError('weights must have the shape of ids')
    T = len(target_slots)
    if isinstance(embedding_func, EmbeddingTable):
        seg, key, key32 = _slot_targets(slots, target_slots, ids, True, id_limit=int(embedding_func.weight.shape[0]))
    else:
        (seg, key), key32 = _slot_targets(slots, target_slots, ids, True), None
    if seg.shape != ids.shape:
        raise ValueError('slots and ids must have the same (B, C) shape')
    srt = _Sorted(key, key32)
    mean = method == 'mean'
    if isinstance(embedding_func, EmbeddingTable):
        return _PoolFunction.apply(embedding_func.weight, ids, seg, weights, T, mean, srt, 'sparse' if embedding_func.sparse_grad else True)
    B, C = ids.shape
    N = B * C
    dev = ids.device
    if not use_unique:
        # the reference embeds every pooled entry separately (:312-313); entries that are not pooled get id 0's row index
        # but weight in no segment.  Done through the unique path with identity inverse = one lookup per entry.
        emb = embedding_func(torch.where(seg.reshape(-1) >= 0, ids.reshape(-1), torch.zeros_like(ids.reshape(-1))))
        rows = torch.arange(N, dtype=torch.int64, device=dev).reshape(B, C)
        srt_id = _Sorted(torch.where(seg >= 0, rows, torch.full_like(rows, KEY_NOT_POOLED)))
        return _PoolFunction.apply(emb, rows, seg, weights, T, mean, srt_id, True)
    s = srt.segments()
    unique = torch.empty(max(N, 1), dtype=torch.int64, device=dev)
    inverse = torch.empty((B, C), dtype=torch.int64, device=dev)
    n_unique = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.call('recnow_embed_unique', _lib.ptr(key), _lib.ptr(s.order), _lib.ptr(s.seg_id), _lib.ptr(s.seg_first), _lib.ptr(s.n_seg), N,
              _lib.ptr(unique), _lib.ptr(inverse), _lib.ptr(n_unique), _lib.stream())
    U = int(n_unique.item())                       # data-dependent size, as tf.unique's output (the one host sync of the path)
    if U < 0:
        s.num_segments()                           # raises: the cooperative grouping kernel timed out (n_seg = -1)
    emb = embedding_func(unique[:U])               # (U, D), ids in ascending order
    if emb.dim() != 2 or emb.shape[0] != U:
        raise ValueError('embedding_func must map n ids to an (n, D) tensor')
    if U == 0:
        emb = emb.new_zeros((1, emb.shape[1] if emb.dim() == 2 else 1))
    return _PoolFunction.apply(emb, inverse, seg, weights, T, mean, srt, False)
