"""GPU parity for the SURVEY section 8f rows: InnerPNNLayer, SENETLayer, attention_by_dot_product, focal_crossentropy_loss
(HIP) against the reference's own goldens / literal test vectors and, forward + backward, against the oracle (fp64 autograd
of the dense restatement).  Tolerance 1e-5 relative (north_star)."""
import os

import numpy as np
import pytest
import torch

import dense_ref as R
from test_oracle_golden import ATTN_DOC, ATTN_GOLDEN, ATTN_USER

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def close(a, b, rtol=RTOL, scale=None):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    assert a.shape == b.shape, (a.shape, b.shape)
    s = max(np.abs(b).max() if scale is None else scale, 1e-30)
    err = np.abs(a - b).max()
    if os.environ.get('RECNOW_TEST_MARGIN_LOG') and err > 0.3 * rtol * s:      # diagnostics: comparisons that use more than 30 % of their bound
        with open(os.environ['RECNOW_TEST_MARGIN_LOG'], 'a') as fh:
            fh.write('%.3f of the bound  %s  %s\n' % (err / (rtol * s), os.environ.get('PYTEST_CURRENT_TEST', ''), ''))
    assert err <= rtol * s, 'max err %.3g vs scale %.3g' % (err, s)


# ---- InnerPNN --------------------------------------------------------------------------------------------------------
def test_inner_pnn_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_inner_pnn_layer.py:18-39
    from rec_now_amd.layers.inner_pnn_layer import InnerPNNLayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('inner_pnn')
    embeddings = [torch.from_numpy(x).to(dev) for x in g['inputs']]
    result = InnerPNNLayer(name='InnerPNNLayer')(embeddings)
    assert result.shape == (4, 3)
    assert calc_sum_of_abs_diff(result, g['golden']) < 1e-5


@pytest.mark.parametrize('B,F,D', [(1, 2, 1), (5, 3, 2), (300, 64, 16), (129, 70, 8), (64, 130, 4), (33, 7, 64), (10, 1, 4),
                                   # the Gram (MFMA) forward: one / two row blocks, every D it takes, ragged F, and more rows than
                                   # waves in the grid (the prefetching row loop)
                                   (257, 33, 12), (131, 40, 8), (100, 32, 4), (50, 17, 16), (77, 63, 12), (3, 2, 4),
                                   (70001, 8, 4), (66000, 34, 8)])
def test_inner_pnn_fwd_bwd_vs_oracle(dev, B, F, D):
    from rec_now_amd.layers.inner_pnn_layer import InnerPNNLayer
    rng = np.random.default_rng(B + 3 * F + D)
    xs = [rng.uniform(-1, 1, (B, D)).astype(np.float32) for _ in range(F)]
    P = F * (F - 1) // 2
    gy = rng.normal(size=(B, P)).astype(np.float32)
    xd = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in xs]
    y = InnerPNNLayer()(xd)
    assert y.shape == (B, P)
    x64 = [torch.from_numpy(x).double().requires_grad_(True) for x in xs]
    if F == 1:
        return
    ry = R.inner_pnn_layer(x64)
    y.backward(torch.from_numpy(gy).to(dev))
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry, scale=float(D))
    gscale = max(float(x.grad.abs().max()) for x in x64)
    for a, b in zip(xd, x64):
        close(a.grad, b.grad, scale=max(gscale, 1.0))


# ---- SENET -----------------------------------------------------------------------------------------------------------
def test_senet_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_senet_layer.py:18-39
    from rec_now_amd.layers.senet_layer import SENETLayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('senet')
    embeddings = [torch.from_numpy(g['input_%d' % i]).to(dev) for i in range(3)]
    senet_layer = SENETLayer(0.3)
    senet_layer(embeddings)                                       # builds
    assert senet_layer.middle_dim == 1
    name = senet_layer.name
    senet_layer.set_weights_by_name({f'{name}/senet/dense_{i}/{w}': g['dense_%d_%s' % (i, w)] for i in (0, 1) for w in ('kernel', 'bias')})
    result = senet_layer(embeddings)
    assert result.shape == (2, 6)
    assert calc_sum_of_abs_diff(result, g['golden']) < 1e-5


@pytest.mark.parametrize('B,dims,ratio,use_bias', [(7, [1, 2, 3], 0.3, True), (300, [16] * 24, 0.25, True), (65, [4, 8, 4, 32, 1], 0.9, False),
                                                   (1, [5], 1.0, True),
                                                   # the one-pass kernels: no bias, rows % 4 != 0, widest / narrowest shapes they take, one
                                                   # field, hidden width 1; and shapes just outside (F > 64, F*D/4 > 256) on the unfused path
                                                   (1001, [16] * 64, 1.0, False), (6, [4] * 64, 0.5, True), (130, [64] * 16, 0.1, True),
                                                   (9, [8], 1.0, True), (33, [4] * 3, 0.2, False), (50, [4] * 65, 0.5, True), (20, [32] * 40, 0.25, True)])
def test_senet_fwd_bwd_vs_oracle(dev, B, dims, ratio, use_bias, act_inner='relu'):
    from rec_now_amd.layers.senet_layer import SENETLayer
    rng = np.random.default_rng(B + len(dims))
    xs = [rng.uniform(-1, 1, (B, d)).astype(np.float32) for d in dims]
    total = sum(dims)
    gy = rng.normal(size=(B, total)).astype(np.float32)
    layer = SENETLayer(ratio, activation_inner=act_inner, activation_outer='sigmoid', use_bias=use_bias, bias_initializer='random_normal')
    xd = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in xs]
    y = layer(xd)
    y.backward(torch.from_numpy(gy).to(dev))
    w = layer.named_weights()
    name = layer.name
    k = [w[f'{name}/senet/dense_{i}/kernel'] for i in (0, 1)]
    bs = [w[f'{name}/senet/dense_{i}/bias'] for i in (0, 1)] if use_bias else None
    k64 = [t.detach().cpu().double().requires_grad_(True) for t in k]
    b64 = [t.detach().cpu().double().requires_grad_(True) for t in bs] if use_bias else None
    x64 = [torch.from_numpy(x).double().requires_grad_(True) for x in xs]
    ry = R.senet_layer(x64, k64, b64, act_inner, 'sigmoid')
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry)
    for a, b in zip(xd, x64):
        close(a.grad, b.grad, scale=max(float(b.grad.abs().max()), 1.0))
    for a, b in zip(k, k64):
        close(a.grad, b.grad, scale=max(float(b.grad.abs().max()), 1.0))
    if use_bias:
        for a, b in zip(bs, b64):
            close(a.grad, b.grad, scale=max(float(b.grad.abs().max()), 1.0))


# ---- attention_by_dot_product -------------------------------------------------------------------------------------------
def test_attention_by_dot_product_reference_literals(dev):
    # /root/reference/tests/rec_block/test_attention.py:19-56
    from rec_now_amd.rec_block.attention import attention_by_dot_product
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    user_emb = torch.tensor(ATTN_USER, device=dev)
    doc_emb = torch.tensor(ATTN_DOC, device=dev)
    for filter_neg, (true_attn_mat, true_attn_score) in ATTN_GOLDEN.items():
        attn_mat, attn_score = attention_by_dot_product(user_emb, doc_emb, filter_neg=filter_neg)
        assert calc_sum_of_abs_diff(attn_mat, true_attn_mat) < 1e-5
        assert calc_sum_of_abs_diff(attn_score, true_attn_score) < 1e-5


@pytest.mark.parametrize('B,L,D,filter_neg', [(1, 1, 1, False), (33, 7, 16, True), (257, 50, 64, False), (5, 3, 200, True), (4, 0, 8, False),
                                             # the float4 kernels: every lanes-per-position count, ragged piece counts, rows % 4 != 0
                                             (131, 50, 16, True), (66, 5, 4, False), (39, 9, 8, True), (70, 11, 32, False), (7, 1, 16, True),
                                             (1026, 65, 16, False), (35, 6, 12, True)])
def test_attention_by_dot_product_fwd_bwd_vs_oracle(dev, B, L, D, filter_neg):
    from rec_now_amd.rec_block.attention import attention_by_dot_product
    rng = np.random.default_rng(B + L + D)
    u = rng.uniform(-1, 1, (B, L, D)).astype(np.float32)
    d = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    gm = rng.normal(size=(B, D)).astype(np.float32)
    gs = rng.normal(size=(B, 1)).astype(np.float32)
    ud, dd = (torch.from_numpy(v).to(dev).requires_grad_(True) for v in (u, d))
    mat, ssum = attention_by_dot_product(ud, dd, filter_neg=filter_neg)
    (mat * torch.from_numpy(gm).to(dev)).sum().add((ssum * torch.from_numpy(gs).to(dev)).sum()).backward()
    u64, d64 = (torch.from_numpy(v).double().requires_grad_(True) for v in (u, d))
    rmat, rsum = R.attention_by_dot_product(u64, d64, filter_neg=filter_neg)
    ((rmat * torch.from_numpy(gm).double()).sum() + (rsum * torch.from_numpy(gs).double()).sum()).backward()
    sc = float(D * max(L, 1))
    close(mat, rmat, scale=sc)
    close(ssum, rsum, scale=sc)
    if L > 0:
        close(ud.grad, u64.grad, scale=max(float(u64.grad.abs().max()), 1.0) * 4)
    close(dd.grad, d64.grad, scale=max(float(d64.grad.abs().max()), 1.0) * 4)


# ---- focal loss ------------------------------------------------------------------------------------------------------
def test_focal_crossentropy_reference_literals(dev):
    # /root/reference/tests/rec_block/test_focal_loss.py:17-28
    from rec_now_amd.rec_block.focal_loss import focal_crossentropy_loss
    labels = torch.tensor([1, 1, 0, 0], dtype=torch.float32, device=dev).reshape(-1, 1)
    logits = torch.tensor([0.9, 0.8, 0.7, 0.6], dtype=torch.float32, device=dev).reshape(-1, 1)
    assert abs(float(focal_crossentropy_loss(labels, logits, alpha=None, gamma=None, return_mean=True)) - 0.71323216) < 1e-5
    assert abs(float(focal_crossentropy_loss(labels, logits, alpha=0.25, gamma=None, return_mean=True)) - 0.44589227) < 1e-5
    assert abs(float(focal_crossentropy_loss(labels, logits, alpha=None, gamma=1, return_mean=True)) - 0.40516436) < 1e-5


def test_focal_crossentropy_argument_errors(dev):
    from rec_now_amd.rec_block.focal_loss import focal_crossentropy_loss
    z = torch.zeros(4, device=dev)
    with pytest.raises(ValueError):
        focal_crossentropy_loss(z, z, alpha=1.5)
    with pytest.raises(ValueError):
        focal_crossentropy_loss(z, z, gamma=-1.0)


@pytest.mark.parametrize('B', [1, 1000, 70001])
@pytest.mark.parametrize('alpha,gamma,stop_w,mean', [(0.25, 2.0, False, True), (None, 0.5, False, False), (0.7, None, False, True),
                                                     (0.25, 3.0, True, False), (None, None, False, True)])
def test_focal_crossentropy_fwd_bwd_vs_oracle(dev, B, alpha, gamma, stop_w, mean):
    from rec_now_amd.rec_block.focal_loss import focal_crossentropy_loss
    rng = np.random.default_rng(B)
    z = (rng.random(B) < 0.3).astype(np.float32)
    x = rng.normal(0, 3, B).astype(np.float32)
    g = rng.normal(size=() if mean else (B,)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = focal_crossentropy_loss(torch.from_numpy(z).to(dev), xd, alpha=alpha, gamma=gamma, stop_weight_gradient=stop_w, return_mean=mean)
    (out * torch.from_numpy(g).to(dev)).sum().backward()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    ref = R.focal_crossentropy_loss(torch.from_numpy(z).double(), x64, alpha=alpha, gamma=gamma, stop_weight_gradient=stop_w, return_mean=mean)
    (ref * torch.from_numpy(g).double()).sum().backward()
    close(out, ref, rtol=2e-5 if gamma and gamma < 1 else RTOL)
    close(xd.grad, x64.grad, rtol=2e-5, scale=max(float(x64.grad.abs().max()), 1e-12))


def test_empty_batches(dev):
    """B = 0 through every widened op, forward and backward (the reference's ops accept empty batches too)."""
    from rec_now_amd.layers.inner_pnn_layer import InnerPNNLayer
    from rec_now_amd.layers.senet_layer import SENETLayer
    from rec_now_amd.rec_block.attention import attention_by_dot_product
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    from rec_now_amd.rec_block.focal_loss import focal_crossentropy_loss
    z = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)       # noqa: E731
    out = InnerPNNLayer()([z(0, 4) for _ in range(3)])
    assert out.shape == (0, 3)
    out.sum().backward()
    out = SENETLayer(0.5)([z(0, 4) for _ in range(3)])
    assert out.shape == (0, 12)
    out.sum().backward()
    mat, ssum = attention_by_dot_product(z(0, 3, 4), z(0, 4))
    assert mat.shape == (0, 4) and ssum.shape == (0, 1)
    mat.sum().backward()
    assert focal_crossentropy_loss(torch.zeros(0, device=dev), z(0), return_mean=False).shape == (0,)
    assert torch.isnan(focal_crossentropy_loss(torch.zeros(0, device=dev), z(0)))          # tf.reduce_mean of nothing
    table = EmbeddingTable(torch.ones(5, 2, device=dev))
    ids = torch.zeros(0, 3, dtype=torch.int64, device=dev)
    out = embedding_using_sparse_batch_segment_ids(table, ids.to(torch.int32), [1], ids)
    assert out.shape == (0, 1, 2)
    out.sum().backward()
    assert embedding_using_sparse_batch_segment_ids(table, torch.zeros(2, 3, dtype=torch.int32, device=dev), [], torch.zeros(2, 3, dtype=torch.int64, device=dev)).shape == (2, 0, 2)


def test_inner_pnn_and_senet_bitwise_reproducible(dev):
    """Fixed summation order everywhere (MFMA accumulation order, LDS hand-overs inside one wave): identical bits per run."""
    from rec_now_amd.layers.inner_pnn_layer import InnerPNNLayer
    from rec_now_amd.layers.senet_layer import SENETLayer
    rng = np.random.default_rng(3)
    B, F, D = 5000, 64, 16
    xs = [torch.from_numpy(rng.normal(size=(B, D)).astype(np.float32)).to(dev) for _ in range(F)]
    gp = torch.from_numpy(rng.normal(size=(B, F * (F - 1) // 2)).astype(np.float32)).to(dev)
    gs = torch.from_numpy(rng.normal(size=(B, F * D)).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    senet = SENETLayer(0.5)
    runs = []
    for _ in range(3):
        leaves = [x.clone().requires_grad_(True) for x in xs]
        y = InnerPNNLayer()(leaves)
        y.backward(gp)
        z = SENETLayer.__call__(senet, [x.clone().requires_grad_(True) for x in xs])
        runs.append((y.detach().clone(), torch.stack([x.grad for x in leaves]), z.detach().clone()))
    for y, g, z in runs[1:]:
        assert torch.equal(y, runs[0][0]) and torch.equal(g, runs[0][1]) and torch.equal(z, runs[0][2])
