"""GPU parity for the pooled embedding lookup: rec_block/embedding_util.py against the reference's
own test vectors (bit-exact: the vectors are small integers) and, forward + backward, against the oracle."""
import numpy as np
import pytest
import torch

import dense_ref as R
from test_oracle_golden import embed_pool_case

pytestmark = pytest.mark.gpu
T_ = torch.from_numpy


def test_sparse_batch_segment_ids_of_targets_reference_vector(dev):
    # /root/reference/tests/rec_block/test_embedding_util.py:55-69 (integer path: exact)
    from rec_now_amd.rec_block.embedding_util import sparse_batch_segment_ids_of_targets
    slots = torch.tensor([[0, 1, 1, 2, 3, 3], [1, 3, 3, 2, 5, 5]], device=dev)
    mask, sp_segment_ids, num_rows, num_ids, num_segments = sparse_batch_segment_ids_of_targets(slots, [1, 3, 5])
    assert mask.cpu().tolist() == [[False, True, True, False, True, True], [True, True, True, False, True, True]]
    assert sp_segment_ids.cpu().tolist() == [0, 0, 1, 1, 3, 4, 4, 5, 5]
    assert (num_rows, num_ids, num_segments) == (2, 3, 6)

@pytest.mark.parametrize('T,n_slots,dtype', [(1, 4, np.int32), (3, 9, np.int32), (64, 80, np.int32), (66, 70, np.int64), (5000, 6000, np.int64),
                                             (4097, 5000, np.int32), (0, 5, np.int32)])
def test_slot_target_map_exact(dev, T, n_slots, dtype):
    """seg[b][c] = index of slots[b][c] in target_slots or -1 (embedding_util.py:138-195), and the sort key: the id of pooled
    entries, KEY_NOT_POOLED otherwise.  Covers target lists that are not a multiple of 4 and longer than one LDS piece."""
    from rec_now_amd.rec_block import embedding_util as E
    rng = np.random.default_rng(T + n_slots)
    slots = rng.integers(0, n_slots, (37, 29)).astype(dtype)
    if dtype == np.int64:
        slots[0, :5] = (1 << 40) + np.arange(5)                   # wide slot values
    targets = [int(v) for v in rng.permutation(n_slots)[:T]]
    if dtype == np.int64 and T > 2:
        targets[1] = (1 << 40) + 3
    ids = rng.integers(0, 1 << 40, slots.shape).astype(np.int64)
    seg, key = E._slot_targets(T_(slots).to(dev), targets, T_(ids).to(dev), True)
    pos = {v: i for i, v in enumerate(targets)}
    exp = np.array([[pos.get(int(v), -1) for v in row] for row in slots], np.int32)
    assert np.array_equal(seg.cpu().numpy(), exp)
    assert np.array_equal(key.cpu().numpy(), np.where(exp >= 0, ids, E.KEY_NOT_POOLED))


@pytest.mark.parametrize('path', ['table', 'callable_unique', 'callable_no_unique'])
def test_embedding_using_sparse_batch_segment_ids_reference_vectors(dev, path):
    # /root/reference/tests/rec_block/test_embedding_util.py:71-109
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    params, ids, slots, target_slots, weights, exp_w, exp_n = embed_pool_case()
    pd = T_(params).to(dev)
    if path == 'table':
        embedding_func = EmbeddingTable(pd)
    else:
        def embedding_func(i): return pd[i]
    kw = dict(use_unique=(path != 'callable_no_unique'))
    out = embedding_using_sparse_batch_segment_ids(embedding_func, T_(slots).to(dev), target_slots, T_(ids).to(dev),
                                                   weights=T_(weights).to(dev), **kw)
    assert np.array_equal(out.detach().cpu().numpy(), np.array(exp_w, np.float32))
    out = embedding_using_sparse_batch_segment_ids(embedding_func, T_(slots).to(dev), target_slots, T_(ids).to(dev), weights=None, **kw)
    assert np.array_equal(out.detach().cpu().numpy(), np.array(exp_n, np.float32))


@pytest.mark.parametrize('B,C,T,D,V,method,use_w', [(1, 1, 1, 1, 3, 'sum', False), (37, 20, 5, 16, 50, 'sum', True), (300, 64, 24, 16, 1000, 'mean', True),
                                                    (64, 33, 3, 70, 7, 'mean', False), (5, 8, 4, 8, 100, 'sum', True),
                                                    # more than 64 id columns (two metadata rounds), D = 32 / 12 lane groups
                                                    (40, 130, 9, 32, 300, 'sum', True), (33, 65, 70, 12, 40, 'mean', False)])
@pytest.mark.parametrize('path', ['table', 'callable_unique', 'callable_no_unique'])
def test_embedding_pool_fwd_bwd_vs_oracle(dev, B, C, T, D, V, method, use_w, path):
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    rng = np.random.default_rng(B * 7 + C + T + D)
    n_slots = T + 3
    slots = rng.integers(0, n_slots, (B, C)).astype(np.int32)
    target_slots = [int(v) for v in rng.permutation(n_slots)[:T]]
    ids = rng.integers(0, V, (B, C)).astype(np.int64)
    if B > 4:
        ids[: B // 2, : max(C // 2, 1)] = 1 % V                      # a hot id with many entries
    weights = rng.uniform(0.5, 1.5, (B, C)).astype(np.float32) if use_w else None
    params = rng.normal(size=(V, D)).astype(np.float32)
    gout = rng.normal(size=(B, T, D)).astype(np.float32)
    table = torch.nn.Parameter(T_(params).to(dev))
    if path == 'table':
        embedding_func = EmbeddingTable(table)
    else:
        from rec_now_amd.rec_block.embedding_util import EmbeddingTable as ET
        lookup = ET(table)
        def embedding_func(i): return lookup(i)                 # a user callable (here: the HIP row lookup)
    out = embedding_using_sparse_batch_segment_ids(embedding_func, T_(slots).to(dev), target_slots, T_(ids).to(dev),
                                                   weights=None if weights is None else T_(weights).to(dev), method=method,
                                                   use_unique=(path != 'callable_no_unique'))
    out.backward(T_(gout).to(dev))
    p64 = T_(params).double().requires_grad_(True)
    ref = R.embedding_using_sparse_batch_segment_ids(p64, T_(slots), target_slots, T_(ids), None if weights is None else T_(weights).double(), method)
    ref.backward(T_(gout).double())
    scale = max(float(ref.detach().abs().max()), 1.0)
    assert np.abs(out.detach().cpu().double().numpy() - ref.detach().numpy()).max() <= 1e-5 * scale
    gs = max(float(p64.grad.abs().max()), 1.0)
    assert np.abs(table.grad.cpu().double().numpy() - p64.grad.numpy()).max() <= 1e-5 * gs


def test_embedding_pool_no_target_present_and_errors(dev):
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    table = EmbeddingTable(torch.ones(4, 2, device=dev))
    ids = torch.tensor([[0, 1], [2, 3]], device=dev)
    slots = torch.tensor([[7, 7], [7, 7]], device=dev)
    out = embedding_using_sparse_batch_segment_ids(table, slots, [1, 2], ids, method='mean')
    assert out.shape == (2, 2, 2) and float(out.abs().sum()) == 0.0
    out = embedding_using_sparse_batch_segment_ids(lambda i: torch.ones(i.numel(), 2, device=dev), slots, [1, 2], ids)
    assert out.shape == (2, 2, 2) and float(out.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        embedding_using_sparse_batch_segment_ids(table, slots, [1, 1], ids)
    with pytest.raises(ValueError):
        embedding_using_sparse_batch_segment_ids(table, slots, [1], ids, method='max')


def test_embedding_pool_gradient_is_bitwise_reproducible(dev):
    """No float atomics anywhere on the path: the table gradient of the pooled lookup is bit-identical from run to run,
    including a hot id whose entries span hundreds of 32-entry chunks (the workgroup join) and ids with a handful of entries
    (the lane-group join)."""
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    rng = np.random.default_rng(5)
    B, C, T, D, V = 3000, 40, 12, 16, 5000
    slots = T_(rng.integers(0, T + 2, (B, C)).astype(np.int32)).to(dev)
    ids = (rng.zipf(1.3, (B, C)) % V).astype(np.int64)
    ids[:, :8] = 7                                                  # 24 000 entries of one id
    ids = T_(ids).to(dev)
    w = T_(rng.uniform(0.5, 1.5, (B, C)).astype(np.float32)).to(dev)
    gy = T_(rng.normal(size=(B, T, D)).astype(np.float32)).to(dev)
    params = T_(rng.normal(size=(V, D)).astype(np.float32)).to(dev)
    grads = []
    for _ in range(3):
        table = EmbeddingTable(torch.nn.Parameter(params.clone()))
        out = embedding_using_sparse_batch_segment_ids(table, slots, list(range(T)), ids, weights=w, method='mean')
        out.backward(gy)
        grads.append((out.detach().clone(), table.weight.grad.clone()))
    for o, g in grads[1:]:
        assert torch.equal(o, grads[0][0]) and torch.equal(g, grads[0][1])
    assert float(grads[0][1][7].abs().sum()) > 0


@pytest.mark.parametrize('B,C,T,D,V,method', [(37, 20, 5, 16, 50, 'sum'), (300, 64, 24, 16, 1000, 'mean'), (64, 33, 3, 70, 7, 'mean'),
                                              (40, 130, 9, 32, 300, 'sum'), (9, 7, 2, 3, 11, 'mean')])
@pytest.mark.parametrize('path', ['table', 'callable_unique'])
def test_embedding_pool_gradient_wrt_id_weights(dev, B, C, T, D, V, method, path):
    """d out / d weights, as TF autodiff gives it through `embeddings * expand_dims(sp_weights, -1)`
    (/root/reference/rec_now/rec_block/embedding_util.py:315-317), together with the table gradient."""
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    rng = np.random.default_rng(B + C * 3 + T + D)
    n_slots = T + 2
    slots = rng.integers(0, n_slots, (B, C)).astype(np.int32)
    target_slots = [int(v) for v in rng.permutation(n_slots)[:T]]
    ids = rng.integers(0, V, (B, C)).astype(np.int64)
    weights = rng.uniform(-1.5, 1.5, (B, C)).astype(np.float32)
    params = rng.normal(size=(V, D)).astype(np.float32)
    gout = rng.normal(size=(B, T, D)).astype(np.float32)
    table = torch.nn.Parameter(T_(params).to(dev))
    lookup = EmbeddingTable(table)
    embedding_func = lookup if path == 'table' else (lambda i: lookup(i))
    wd = T_(weights).to(dev).requires_grad_(True)
    out = embedding_using_sparse_batch_segment_ids(embedding_func, T_(slots).to(dev), target_slots, T_(ids).to(dev), weights=wd, method=method)
    out.backward(T_(gout).to(dev))
    p64 = T_(params).double().requires_grad_(True)
    w64 = T_(weights).double().requires_grad_(True)
    ref = R.embedding_using_sparse_batch_segment_ids(p64, T_(slots), target_slots, T_(ids), w64, method)
    ref.backward(T_(gout).double())
    ws = max(float(w64.grad.abs().max()), 1e-30)
    assert np.abs(wd.grad.cpu().double().numpy() - w64.grad.numpy()).max() <= 1e-5 * ws
    gs = max(float(p64.grad.abs().max()), 1.0)
    assert np.abs(table.grad.cpu().double().numpy() - p64.grad.numpy()).max() <= 1e-5 * gs
    # weights only (a frozen table): no table gradient is formed
    frozen = EmbeddingTable(torch.nn.Parameter(T_(params).to(dev), requires_grad=False))
    wd2 = T_(weights).to(dev).requires_grad_(True)
    embedding_using_sparse_batch_segment_ids(frozen, T_(slots).to(dev), target_slots, T_(ids).to(dev), weights=wd2,
                                             method=method).backward(T_(gout).to(dev))
    assert torch.equal(wd2.grad, wd.grad) if path == 'table' else True
    assert frozen.weight.grad is None


def test_embedding_ids_outside_the_table_read_as_zero_rows(dev):
    """An id < 0 or >= V must not read outside the table: it contributes a zero row in the forward (tf.nn.embedding_lookup on a
    GPU returns zeros), is dropped by the backward's scatter, and still counts as a pooled entry for 'mean'."""
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    V, D = 6, 4
    params = np.arange(V * D, dtype=np.float32).reshape(V, D) + 1.0
    ids = np.array([[0, 7, 5, -3], [6, 1, 1, 2]], dtype=np.int64)
    slots = np.array([[1, 1, 2, 2], [1, 1, 2, 2]], dtype=np.int32)
    for method in ('sum', 'mean'):
        table = EmbeddingTable(torch.nn.Parameter(T_(params).to(dev)))
        wd = torch.ones(2, 4, device=dev, requires_grad=True)
        out = embedding_using_sparse_batch_segment_ids(table, T_(slots).to(dev), [1, 2], T_(ids).to(dev), weights=wd, method=method)
        out.sum().backward()
        ok = (ids >= 0) & (ids < V)
        rows = np.where(ok[:, :, None], params[np.clip(ids, 0, V - 1)], 0.0)
        ref = np.stack([rows[:, :2].sum(1), rows[:, 2:].sum(1)], axis=1) / (2.0 if method == 'mean' else 1.0)
        assert np.abs(out.detach().cpu().numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
        dt = np.zeros((V, D))
        for b in range(2):
            for c in range(4):
                if ok[b, c]:
                    dt[ids[b, c]] += 1.0 / (2.0 if method == 'mean' else 1.0)
        assert np.abs(table.weight.grad.cpu().numpy() - dt).max() <= 1e-6
        dw = np.where(ok, rows.sum(2), 0.0) / (2.0 if method == 'mean' else 1.0)
        assert np.abs(wd.grad.cpu().numpy() - dw).max() <= 1e-5 * np.abs(dw).max()
    # plain lookup through EmbeddingTable(ids)
    got = EmbeddingTable(T_(params).to(dev))(torch.tensor([2, 9, -1], device=dev))
    assert torch.equal(got[0].cpu(), T_(params[2])) and float(got[1:].abs().sum()) == 0.0


def test_embedding_table_sparse_gradient(dev):
    """EmbeddingTable(sparse_grad=True): the table gradient as a sparse COO tensor of the looked-up rows equals the dense one."""
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    rng = np.random.default_rng(21)
    B, C, T, D, V = 200, 30, 6, 16, 500
    slots = T_(rng.integers(0, T + 2, (B, C)).astype(np.int32)).to(dev)
    ids = rng.integers(0, V, (B, C)).astype(np.int64)
    ids[3, 4], ids[7, 1] = V + 5, -2                                   # ids outside the table are dropped
    ids = T_(ids).to(dev)
    w = T_(rng.uniform(0.5, 1.5, (B, C)).astype(np.float32)).to(dev)
    gy = T_(rng.normal(size=(B, T, D)).astype(np.float32)).to(dev)
    params = T_(rng.normal(size=(V, D)).astype(np.float32)).to(dev)
    grads = []
    for sparse in (False, True):
        table = EmbeddingTable(torch.nn.Parameter(params.clone()), sparse_grad=sparse)
        out = embedding_using_sparse_batch_segment_ids(table, slots, list(range(T)), ids, weights=w, method='mean')
        out.backward(gy)
        g = table.weight.grad
        assert g.is_sparse == sparse
        grads.append(g.to_dense() if sparse else g)
    assert torch.equal(grads[0], grads[1])
    assert grads[1].abs().sum() > 0
