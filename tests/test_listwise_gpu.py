"""GPU parity: rec_now_amd.rec_block.listwise_loss_from_batch vs the reference goldens and the oracle.
Reads like /root/reference/tests/rec_block/test_listwise_loss_from_batch.py plus randomized / edge cases."""
import numpy as np
import pytest
import torch

import dense_ref as R

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _mod():
    from rec_now_amd.rec_block import listwise_loss_from_batch as M
    return M


def test_listwise_loss(dev):
    # reference test_listwise_loss: two valid lists
    M = _mod()
    sample_group_idx_var = torch.tensor([[1, 1, 2, 1, 2, 2, 3, 4.]], device=dev).t()
    labels = torch.tensor([[1, 1, 1, 0, 0, 0, 1, 0.]], device=dev).t()
    logits = torch.tensor([[0.1, 0.01, 0.2, 0.001, 0.02, 0.002, 0.3, 0.4]], device=dev).t()
    sample_mask, labels_for_softmax, logits_for_softmax = M.to_listwise_sample(sample_group_idx_var, labels, logits)
    n_valid_list = labels_for_softmax.shape[0]
    n_sample_per_valid_group = M.nan_to_zero(sample_mask.float().sum(-1).mean())
    listwise_loss = M.listwise_loss_via_softmax_cross_entropy_with_logits(labels_for_softmax=labels_for_softmax,
                                                                          logits_for_softmax=logits_for_softmax)
    assert n_valid_list == 2
    assert abs(n_sample_per_valid_group.item() - 3.0) < 1e-6
    assert abs(listwise_loss.item() - 1.0291535) < 1e-4
    assert abs(M.listwise_loss_from_batch(sample_group_idx_var, labels, logits).item() - 1.0291535) < 1e-4
    # dense outputs equal the oracle's, element for element
    rm, rl, rz = R.to_listwise_sample(sample_group_idx_var.cpu(), labels.cpu(), logits.cpu())
    assert np.array_equal(sample_mask.cpu().numpy(), rm.numpy())
    assert np.allclose(labels_for_softmax.cpu().numpy(), rl.numpy(), rtol=1e-6)
    assert np.allclose(logits_for_softmax.cpu().numpy(), rz.numpy(), rtol=1e-6)


def test_listwise_loss_case2(dev):
    # reference test_listwise_loss_case2: no valid list -> loss 0 (NaN -> 0)
    M = _mod()
    g = torch.tensor([[3, 4.]], device=dev).t()
    labels = torch.tensor([[1, 0.]], device=dev).t()
    logits = torch.tensor([[0.3, 0.4]], device=dev).t()
    sample_mask, labels_for_softmax, logits_for_softmax = M.to_listwise_sample(g, labels, logits)
    assert labels_for_softmax.shape == (0, 2)
    n = M.nan_to_zero(sample_mask.float().sum(-1).mean())
    assert n.item() == 0.0
    loss = M.listwise_loss_via_softmax_cross_entropy_with_logits(labels_for_softmax=labels_for_softmax, logits_for_softmax=logits_for_softmax)
    assert abs(loss.item()) < 1e-4
    lg = logits.clone().requires_grad_(True)
    fused, nv = M.listwise_loss_from_batch(g, labels, lg, return_num_list=True)
    fused.backward()
    assert fused.item() == 0.0 and nv.item() == 0.0 and float(lg.grad.abs().max()) == 0.0


def test_nan_to_zero_and_row_helpers(dev):
    M = _mod()
    with pytest.raises(ValueError):
        M.nan_to_zero(torch.zeros(2, device=dev))
    assert M.nan_to_zero(torch.tensor(float('nan'), device=dev)).item() == 0.0
    assert M.nan_to_zero(torch.tensor(2.5, device=dev)).item() == 2.5
    x = torch.tensor([[0., 0.], [0., 2.], [-1., 0.]], device=dev)
    assert M.row_not_all_zero(x).cpu().tolist() == [False, True, True]
    assert M.row_has_value_greater_than(x, 0.5).cpu().tolist() == [False, True, False]
    assert M.row_has_value_less_than(x, 0.0).cpu().tolist() == [False, False, True]


def _case(B, G, seed, levels=2):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, G, B).astype(np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32) if levels == 2 else rng.integers(0, levels, B).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    return g, y, s


@pytest.mark.parametrize('B,G,seed', [(8, 3, 0), (200, 9, 1), (1000, 40, 2), (1500, 1, 3), (300, 300, 4)])
@pytest.mark.parametrize('do_mask', [True, False])
def test_fused_and_dense_paths_vs_oracle(dev, B, G, seed, do_mask):
    M = _mod()
    g, y, s = _case(B, G, seed, levels=3)
    rng = np.random.default_rng(seed + 50)
    gd, yd = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev)
    # oracle (fp64)
    s64 = torch.from_numpy(s).double().requires_grad_(True)
    rm, rl, rz = R.to_listwise_sample(torch.from_numpy(g), torch.from_numpy(y).double(), s64, do_mask_logits=do_mask,
                                      value_of_masked_logit=-1e9, pos_neg_th=0.5)
    gv = rl.shape[0]
    w = rng.uniform(0.5, 2.0, gv).astype(np.float32)
    rloss = R.listwise_loss_via_softmax_cross_entropy_with_logits(rl, rz, torch.from_numpy(w).double())
    if rloss.requires_grad:           # with no valid list the oracle's loss is the constant 0
        rloss.backward()
    ref_grad = s64.grad.numpy() if s64.grad is not None else np.zeros(B)
    scale = max(np.abs(ref_grad).max(), 1e-12)
    # fused path
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, nv = M.listwise_loss_from_batch(gd, yd, sd, weights=torch.from_numpy(w).to(dev), do_mask_logits=do_mask, return_num_list=True)
    loss.backward()
    assert int(nv.item()) == gv
    assert abs(loss.item() - rloss.item()) <= RTOL * max(1.0, abs(rloss.item()))
    assert np.abs(sd.grad.cpu().numpy() - ref_grad).max() <= RTOL * scale
    # dense path (reference API)
    sd2 = torch.from_numpy(s).to(dev).requires_grad_(True)
    m, lab, lg = M.to_listwise_sample(gd, yd, sd2, do_mask_logits=do_mask)
    assert np.array_equal(m.cpu().numpy(), rm.numpy())
    assert lab.shape == rl.shape
    if gv > 0:
        assert np.abs(lab.cpu().numpy() - rl.numpy()).max() <= 1e-6
    loss2 = M.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg, torch.from_numpy(w).to(dev))
    loss2.backward()
    assert abs(loss2.item() - rloss.item()) <= RTOL * max(1.0, abs(rloss.item()))
    assert np.abs(sd2.grad.cpu().numpy() - ref_grad).max() <= RTOL * scale


def test_do_reduce_false_per_list_losses(dev):
    M = _mod()
    g, y, s = _case(500, 20, 7)
    gd, yd = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev)
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    per = M.listwise_loss_from_batch(gd, yd, sd, do_reduce=False)
    s64 = torch.from_numpy(s).double().requires_grad_(True)
    _, rl, rz = R.to_listwise_sample(torch.from_numpy(g), torch.from_numpy(y).double(), s64)
    rper = R.listwise_loss_via_softmax_cross_entropy_with_logits(rl, rz, do_reduce=False)
    assert per.shape == rper.shape
    assert np.abs(per.detach().cpu().numpy() - rper.detach().numpy()).max() <= RTOL * np.abs(rper.detach().numpy()).max()
    up = np.random.default_rng(1).normal(size=per.shape[0]).astype(np.float32)
    (per * torch.from_numpy(up).to(dev)).sum().backward()
    (rper * torch.from_numpy(up).double()).sum().backward()
    assert np.abs(sd.grad.cpu().numpy() - s64.grad.numpy()).max() <= RTOL * np.abs(s64.grad.numpy()).max()


def test_threshold_semantics_B8(dev):
    # label == th is neither positive nor negative
    M = _mod()
    g = torch.tensor([1., 1., 2., 2.], device=dev)
    y = torch.tensor([0.5, 1.0, 0.5, 0.0], device=dev)
    s = torch.zeros(4, device=dev)
    m, lab, lg = M.to_listwise_sample(g, y, s)
    rm, rl, rz = R.to_listwise_sample(g.cpu(), y.cpu(), s.cpu())
    assert lab.shape == rl.shape == (0, 4)


def test_full_size_properties_config5(dev):
    """B=262144, 4096 groups (BASELINE config 5 size): loss equals an independent per-group numpy computation,
    gradient sums to zero within every valid group, bitwise determinism."""
    M = _mod()
    B, G = 262144, 4096
    g, y, s = _case(B, G, 5)
    gd, yd = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev)
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, nv = M.listwise_loss_from_batch(gd, yd, sd, return_num_list=True)
    loss.backward()
    gi = g.astype(np.int64)
    s64, y64 = s.astype(np.float64), y.astype(np.float64)
    mx = np.full(G, -np.inf)
    np.maximum.at(mx, gi, s64)
    z = np.bincount(gi, weights=np.exp(s64 - mx[gi]), minlength=G)
    ysum = np.bincount(gi, weights=y64, minlength=G)
    cnt = np.bincount(gi, minlength=G)
    valid = (ysum > 0) & (ysum < cnt)
    dot = np.bincount(gi, weights=y64 * s64, minlength=G)
    lg = (mx + np.log(z)) - dot / np.where(valid, ysum, 1.0)
    ref = lg[valid].mean()
    assert int(nv.item()) == int(valid.sum())
    assert abs(loss.item() - ref) <= 1e-5 * abs(ref)
    gsum = np.bincount(gi, weights=sd.grad.double().cpu().numpy(), minlength=G)
    assert np.abs(gsum).max() < 1e-8
    # the gradient value by value: d loss / d s_i = (softmax_i - y_i / sum_g y) / G_valid inside a valid list, else 0 --
    # from the per-group fp64 softmax above, no (G, B) tensor needed (listwise_loss_from_batch.py:139-147,166-172)
    gref = np.where(valid[gi], (np.exp(s64 - mx[gi]) / z[gi] - y64 / np.where(valid, ysum, 1.0)[gi]) / valid.sum(), 0.0)
    assert np.abs(sd.grad.double().cpu().numpy() - gref).max() <= RTOL * np.abs(gref).max()
    loss2 = M.listwise_loss_from_batch(gd, yd, sd.detach())
    assert loss2.item() == loss.item()


def test_skewed_group_sizes_vs_oracle(dev):
    """Zipf(1.2) list lengths capped at 2048 (one list spans many waves' worth of rows, most lists are singletons and
    therefore invalid): fused loss / gradient against the dense fp64 oracle."""
    M = _mod()
    rng = np.random.default_rng(9)
    B = 6000
    sizes = []
    while sum(sizes) < B:
        sizes.append(int(min(rng.zipf(1.2), 2048, B - sum(sizes))))
    g = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(g)
    g = g.astype(np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    s64 = torch.from_numpy(s).double().requires_grad_(True)
    rm, rl, rz = R.to_listwise_sample(torch.from_numpy(g), torch.from_numpy(y).double(), s64)
    rloss = R.listwise_loss_via_softmax_cross_entropy_with_logits(rl, rz)
    rloss.backward()
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, nv = M.listwise_loss_from_batch(torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev), sd, return_num_list=True)
    loss.backward()
    assert int(nv.item()) == rl.shape[0] > 0
    assert abs(loss.item() - rloss.item()) <= RTOL * abs(rloss.item())
    assert np.abs(sd.grad.cpu().numpy() - s64.grad.numpy()).max() <= RTOL * np.abs(s64.grad.numpy()).max()


def test_empty_batch(dev):
    """B = 0: no list at all -> the mean over zero lists is NaN in the reference and nan_to_zero turns it into 0."""
    M = _mod()
    z = torch.zeros(0, device=dev)
    s = torch.zeros(0, device=dev, requires_grad=True)
    loss, nv = M.listwise_loss_from_batch(z, z, s, return_num_list=True)
    assert loss.item() == 0.0 and int(nv.item()) == 0
    m, lab, lg = M.to_listwise_sample(z, z, s.detach())
    assert lab.shape == (0, 0) and lg.shape == (0, 0)
