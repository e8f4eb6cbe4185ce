"""GPU parity of the BASELINE.json model compositions end to end (small batches, forward + backward through every
autograd node) against the oracle:
  configs[3]: embeddings -> CINLayer || FMLayer -> Dense head -> pairwise_loss
  configs[4]: x -> PLELayer (3 tasks) -> per-task heads -> listwise loss on task 0 (+ the other heads summed, so every
              weight gets a gradient)"""
import os

import numpy as np
import pytest
import torch

import dense_ref as R

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
    assert err <= rtol * s, 'max err %.3g vs scale %.3g (rel %.3g)' % (err, s, err / s)


def test_config4_cin_fm_pairwise(dev):
    from rec_now_amd.layers.cin_layer import CINLayer
    from rec_now_amd.layers.fm_layer import FMLayer
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss
    rng = np.random.default_rng(4)
    B, F, D, Hs = 640, 12, 8, [16, 16, 8]
    xs = [rng.normal(0, 0.4, (B, D)).astype(np.float32) for _ in range(F)]
    groups = rng.integers(0, 20, B).astype(np.float32)
    labels = (rng.random(B) < 0.3).astype(np.float32)
    cin, fm, head = CINLayer(Hs), FMLayer(), MultiDenseLayer(1, 1)
    xd = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in xs]
    feat = torch.cat([cin(xd), fm(xd)], dim=1)                        # (B, D + 1)
    scores = head(feat).reshape(-1)
    loss = pairwise_loss(scores, torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev))
    loss.backward()
    w64 = [cin.idx2weight[k].detach().cpu().double().requires_grad_(True) for k in range(1, len(Hs) + 1)]
    hk = head.kernel.detach().cpu().double().requires_grad_(True)
    hb = head.bias.detach().cpu().double().requires_grad_(True)
    x64 = [torch.from_numpy(x).double().requires_grad_(True) for x in xs]
    rfeat = torch.cat([R.cin_layer(x64, w64, F, D, True, True), R.fm_layer(x64)], dim=1)
    rs = R.multi_dense_layer(rfeat, hk, hb).reshape(-1)
    ref = R.pairwise_loss(rs, torch.from_numpy(labels).double(), torch.from_numpy(groups))
    ref.backward()
    close(loss, ref)
    gs = max(float(v.grad.abs().max()) for v in x64)
    for a, b in zip(xd, x64):
        close(a.grad, b.grad, scale=gs)
    for k in range(1, len(Hs) + 1):
        close(cin.idx2weight[k].grad, w64[k - 1].grad)
    close(head.kernel.grad, hk.grad)


def test_config4_model_per_rank_size_vs_oracle(dev):
    """configs[3] at its PER-RANK size (global B = 131072 over 8 ranks): B = 16384, F = 64, D = 16, CINLayer([128, 128, 128]) || FMLayer
    -> MultiDenseLayer(1, 1) head on the (B, 17) features -> pairwise_loss with integers(0, 2048) groups: loss, pair count, scores,
    d loss / d x (all rows, all fields) and every weight gradient against the fp64 oracle evaluated chunk-wise
    (tests/_chunked_oracle.py; layers: /root/reference/rec_now/layers/cin_layer.py:101-110, fm_layer.py:36-42,
    multi_dense_layer.py:88-94) with the O(B^2) pair part in oracle/pairs_oracle.c (rec_block/pairwise_loss_from_batch.py:254-279)."""
    import pairs_oracle as PO
    from _chunked_oracle import run_chunked
    from rec_now_amd.layers.cin_layer import CINLayer
    from rec_now_amd.layers.fm_layer import FMLayer
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss
    rng = np.random.default_rng(44)
    B, F, D, Hs = 16384, 64, 16, [128, 128, 128]
    x = rng.normal(0, 0.3, (B, F * D)).astype(np.float32)
    groups = rng.integers(0, 2048, B).astype(np.float32)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    cin, fm, head = CINLayer(Hs), FMLayer(), MultiDenseLayer(1, 1)
    xd = [torch.from_numpy(np.ascontiguousarray(x[:, f * D:(f + 1) * D])).to(dev).requires_grad_(True) for f in range(F)]
    feat = torch.cat([cin(xd), fm(xd)], dim=1)                        # (B, D + 1)
    head(feat[:8])
    with torch.no_grad():
        head.kernel.mul_(0.05)          # scores of O(1) (the FM term of 64 fields is O(10)): pair terms off the softplus(0) plateau, no saturation
        head.bias.fill_(0.2)
    scores = head(feat).reshape(-1)
    loss, n_pair = pairwise_loss(scores, torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev), return_num_pair=True)
    loss.backward()
    w64 = {k: cin.idx2weight[k].detach().cpu().double().requires_grad_(True) for k in range(1, len(Hs) + 1)}
    w64['hk'] = head.kernel.detach().cpu().double().requires_grad_(True)
    w64['hb'] = head.bias.detach().cpu().double().requires_grad_(True)

    def fwd(xc):
        fields = [xc[:, f * D:(f + 1) * D] for f in range(F)]
        rfeat = torch.cat([R.cin_layer_gemm_form(xc, [w64[k] for k in range(1, len(Hs) + 1)], F, D, True, True), R.fm_layer(fields)], dim=1)
        return R.multi_dense_layer(rfeat, w64['hk'], w64['hb']).reshape(-1)

    xt = torch.from_numpy(x)
    (rs,), _, _ = run_chunked(fwd, xt, None, w64, chunk=512, want_dx=False)
    close(scores, torch.from_numpy(rs))
    assert np.abs(rs).std() > 0.05
    rloss, rds, rP = PO.pairwise_bpr(groups, labels, rs.astype(np.float32))
    assert int(n_pair.item()) == rP and rP > B // 2
    assert abs(rloss - np.log(2.0)) > 1e-3
    close(loss, torch.tensor(rloss, dtype=torch.float64))
    _, rdx, rgrads = run_chunked(fwd, xt, torch.from_numpy(rds), w64, chunk=512)
    close(torch.cat([a.grad for a in xd], dim=1), torch.from_numpy(rdx))
    for k in range(1, len(Hs) + 1):
        close(cin.idx2weight[k].grad, torch.from_numpy(rgrads[k]))
    close(head.kernel.grad, torch.from_numpy(rgrads['hk']))
    close(head.bias.grad, torch.from_numpy(rgrads['hb']), scale=float(np.abs(rds).sum()))      # a sum that cancels to ~0: on the scale of its terms


def test_config5_ple_listwise(dev, golden):
    from rec_now_amd.layers.ple_layer import PLELayer
    from rec_now_amd.rec_block.listwise_loss_from_batch import listwise_loss_from_batch
    from test_layers_gpu import _load_ple
    from test_oracle_golden import ple_layers_from_fixture
    g = golden('ple')                                                  # weights of the reference's PLE test (2 tasks, 3 layers)
    rng = np.random.default_rng(5)
    B = 700
    x = rng.normal(0, 1, (B, 4)).astype(np.float32)
    groups = rng.integers(0, 25, B).astype(np.float32)
    labels = (rng.random(B) < 0.3).astype(np.float32)
    layer = PLELayer(2, [[2, 3], [2, 3], [3, 2]], [4, 3, 2], 1, name='PLE', activation='tanh')
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    layer(xd)
    _load_ple(layer, g)
    outs = layer(xd)
    hw = rng.normal(size=(2, 2)).astype(np.float32)                    # fixed linear heads: logit_t = out_t . hw[t]
    logit0 = outs[0] @ torch.from_numpy(hw[0]).to(dev)
    loss = listwise_loss_from_batch(torch.from_numpy(groups).to(dev), torch.from_numpy(labels).to(dev), logit0)
    total = loss + 0.01 * (outs[1] @ torch.from_numpy(hw[1]).to(dev)).sum()
    total.backward()
    g64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in g.items() if k.startswith('l')}
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    routs = R.ple_layer(x64, ple_layers_from_fixture(g64, to=lambda v: v), [True, False, False], activation='tanh')
    rlogit0 = routs[0] @ torch.from_numpy(hw[0]).double()
    _, rl, rz = R.to_listwise_sample(torch.from_numpy(groups), torch.from_numpy(labels).double(), rlogit0)
    rloss = R.listwise_loss_via_softmax_cross_entropy_with_logits(rl, rz)
    rtotal = rloss + 0.01 * (routs[1] @ torch.from_numpy(hw[1]).double()).sum()
    rtotal.backward()
    close(loss, rloss)
    close(xd.grad, x64.grad)
