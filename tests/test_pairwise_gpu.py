"""GPU parity: rec_now_amd.rec_block.pairwise_loss_from_batch (HIP, through the C ABI) vs the oracle.
Reads like the reference's tests/rec_block/test_pairwise_loss_from_batch.py, plus randomized / edge cases.
Integer outputs (pair indices, counts) must be bit-exact; floats within 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

import dense_ref as R

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _mod():
    from rec_now_amd.rec_block import pairwise_loss_from_batch as M
    return M


def _case(dev):
    g = torch.tensor([[1, 1, 2, 2, 2.]], device=dev).t()
    s = torch.tensor([[0, 1, 2, 3, 4.]], device=dev).t()
    y = torch.tensor([[1.1, 0, 0, 1, 1]], device=dev).t()
    return g, s, y


def test_occurance_power_weight(dev):
    M = _mod()
    ids = [1, 1, 2, 4, 4, 4]
    for e, r in zip([0.5, 0.5, 1., 0.33333334, 0.33333334, 0.33333334], M.occurance_power_weight(ids, power=-1).cpu().numpy()):
        assert abs(e - r) < 1e-4
    for e, r in zip([4., 4., 1., 9., 9., 9.], M.occurance_power_weight(ids, power=2).cpu().numpy()):
        assert abs(e - r) < 1e-4


def test_pairwise_loss_reference_goldens(dev):
    # /root/reference/tests/rec_block/test_pairwise_loss_from_batch.py:33-74
    M = _mod()
    g, s, y = _case(dev)

    def pairwise_loss_func(outputs_pos, outputs_neg, weights):
        return M.bpr_loss_func(outputs_pos, outputs_neg, weights, 1.0)

    loss = M.pairwise_loss(s, y, g, pairwise_loss_func, only_use_wrong_order_pair=False, click_occurance_power=-0.5)
    assert abs(loss.item() - 0.5415076) < 1e-4

    def _label_pair_to_weight_func(label_matrix, label_matrix_transpose, **kwargs):
        return (label_matrix > label_matrix_transpose).to(torch.float32)

    loss = M.pairwise_loss(s, y, g, pairwise_loss_func, only_use_wrong_order_pair=False, click_occurance_power=-0.5,
                           label_pair_to_weight_func=_label_pair_to_weight_func)
    assert abs(loss.item() - 0.5415076) < 1e-4
    mask = torch.tensor([[True, True, False, False, False]], device=dev).t()
    loss = M.pairwise_loss(s, y, g, pairwise_loss_func, only_use_wrong_order_pair=False, click_occurance_power=-0.5, mask=mask)
    assert abs(loss.item() - 1.3132617) < 1e-4
    # fused path (default pairloss_func) gives the same three numbers
    assert abs(M.pairwise_loss(s, y, g, click_occurance_power=-0.5).item() - 0.5415076) < 1e-4
    assert abs(M.pairwise_loss(s, y, g, click_occurance_power=-0.5, mask=mask).item() - 1.3132617) < 1e-4


def _random_case(B, n_groups, seed, label_levels=2, float_scores=True):
    rng = np.random.default_rng(seed)
    groups = rng.integers(0, n_groups, B).astype(np.float32)
    scores = rng.normal(size=B).astype(np.float32)
    if label_levels == 2:
        labels = (rng.random(B) < 0.25).astype(np.float32)
    else:
        labels = rng.integers(0, label_levels, B).astype(np.float32)
    return groups, scores, labels


@pytest.mark.parametrize('B,G,seed', [(1, 1, 0), (2, 1, 1), (64, 4, 2), (257, 7, 3), (2048, 32, 4), (4096, 1, 5), (3000, 3000, 6)])
def test_pair_indices_bit_exact(dev, B, G, seed):
    M = _mod()
    g, s, y = _random_case(B, G, seed, label_levels=3)
    for wrong in (False, True):
        pos, neg = M.pair_indices(torch.from_numpy(s).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev),
                                  only_use_wrong_order_pair=wrong)
        rpos, rneg = R.pair_indices(torch.from_numpy(y), torch.from_numpy(g), wrong, torch.from_numpy(s))
        assert pos.dtype == torch.int32
        assert np.array_equal(pos.cpu().numpy(), rpos.numpy().astype(np.int32))
        assert np.array_equal(neg.cpu().numpy(), rneg.numpy().astype(np.int32))


@pytest.mark.parametrize('B,G,seed', [(64, 4, 10), (1000, 17, 11), (2048, 32, 12), (4096, 2, 13)])
@pytest.mark.parametrize('power', [0.0, -0.5, 1.0])
@pytest.mark.parametrize('wrong', [False, True])
def test_fused_loss_and_grad_vs_oracle(dev, B, G, seed, power, wrong):
    M = _mod()
    g, s, y = _random_case(B, G, seed)
    rng = np.random.default_rng(seed + 100)
    mask = rng.random(B) < 0.8
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, n_pair = M.pairwise_loss(sd, torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev), only_use_wrong_order_pair=wrong,
                                   return_num_pair=True, click_occurance_power=power, mask=torch.from_numpy(mask).to(dev))
    loss.backward()
    s64 = torch.from_numpy(s).double().requires_grad_(True)
    rloss, rn = R.pairwise_loss(s64, torch.from_numpy(y).double(), torch.from_numpy(g), only_use_wrong_order_pair=wrong,
                                return_num_pair=True, click_occurance_power=power, mask=torch.from_numpy(mask))
    rloss.backward()
    assert n_pair.item() == rn                                        # integer path: exact
    assert abs(loss.item() - rloss.item()) <= RTOL * max(1.0, abs(rloss.item()))
    gr = s64.grad.numpy()
    scale = max(np.abs(gr).max(), 1e-12)
    assert np.abs(sd.grad.cpu().numpy() - gr).max() <= RTOL * scale


def test_general_path_custom_callables_vs_oracle(dev):
    M = _mod()
    g, s, y = _random_case(777, 13, 21, label_levels=4)
    wf = lambda a, b, **k: torch.clamp(a - b, min=0.0) * k.get('scale', 1.0)   # noqa: E731
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    lf = lambda p, n, w: M.bpr_loss_func(p, n, w, 0.7)                       # noqa: E731
    loss = M.pairwise_loss(sd, torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev), lf, click_occurance_power=-1.0,
                           label_pair_to_weight_func=wf, scale=2.0)
    loss.backward()
    s64 = torch.from_numpy(s).double().requires_grad_(True)
    rlf = lambda p, n, w: R.bpr_loss_func(p, n, w, 0.7)                      # noqa: E731
    rloss = R.pairwise_loss(s64, torch.from_numpy(y).double(), torch.from_numpy(g), rlf, click_occurance_power=-1.0,
                            label_pair_to_weight_func=wf, scale=2.0)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= RTOL * max(1.0, abs(rloss.item()))
    gr = s64.grad.numpy()
    assert np.abs(sd.grad.cpu().numpy() - gr).max() <= RTOL * max(np.abs(gr).max(), 1e-12)


def test_group_list_and_int_ids(dev):
    M = _mod()
    rng = np.random.default_rng(5)
    B = 1500
    g0 = rng.integers(0, 9, B)
    g1 = rng.integers(0, 3, B)
    s = rng.normal(size=B).astype(np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    sd = torch.from_numpy(s).to(dev)
    # list of groups (AND), groups[0] drives the occurrence weights
    loss = M.pairwise_loss(sd, torch.from_numpy(y).to(dev), [torch.from_numpy(g0.astype(np.float32)).to(dev),
                                                            torch.from_numpy(g1.astype(np.float32)).to(dev)],
                           click_occurance_power=-0.5)
    rloss = R.pairwise_loss(torch.from_numpy(s).double(), torch.from_numpy(y).double(),
                            [torch.from_numpy(g0.astype(np.float32)), torch.from_numpy(g1.astype(np.float32))],
                            click_occurance_power=-0.5)
    assert abs(loss.item() - rloss.item()) <= RTOL * max(1.0, abs(rloss.item()))
    # int64 ids beyond 2^24 compare exactly (floats would collide); int32 too
    big = (g0.astype(np.int64) + (1 << 40))
    pos, neg = M.pair_indices(sd, torch.from_numpy(y).to(dev), torch.from_numpy(big).to(dev))
    rpos, rneg = R.pair_indices(torch.from_numpy(y), torch.from_numpy(g0.astype(np.float32)))
    assert np.array_equal(pos.cpu().numpy(), rpos.numpy()) and np.array_equal(neg.cpu().numpy(), rneg.numpy())


def test_float_group_semantics_B1(dev):
    # -0.0 == +0.0, NaN/inf pair with nobody (g_i - g_j == 0.0 in float)
    M = _mod()
    g = torch.tensor([0.0, -0.0, float('nan'), float('nan'), float('inf'), float('inf'), 5.0, 5.0])
    y = torch.tensor([1., 0., 1., 0., 1., 0., 1., 0.])
    s = torch.zeros(8)
    pos, neg = M.pair_indices(s.to(dev), y.to(dev), g.to(dev))
    rpos, rneg = R.pair_indices(y, g)
    assert pos.cpu().tolist() == rpos.tolist() == [0, 6] and neg.cpu().tolist() == rneg.tolist() == [1, 7]


def test_no_pairs_gives_zero_loss_and_grad(dev):
    M = _mod()
    s = torch.randn(16, device=dev, requires_grad=True)
    y = torch.ones(16, device=dev)
    g = torch.arange(16, device=dev, dtype=torch.float32) // 4
    loss, n = M.pairwise_loss(s, y, g, return_num_pair=True)
    loss.backward()
    assert loss.item() == 0.0 and n.item() == 0.0 and float(s.grad.abs().max()) == 0.0


def test_generate_pair_mask_and_vec_to_matrix_pair(dev):
    M = _mod()
    g = torch.tensor([1, 1, 2, 2, 2.], device=dev)
    m = M.generate_pair_mask(g)
    assert np.array_equal(m.cpu().numpy(), R.generate_pair_mask(g.cpu()).numpy())
    mb = M.generate_pair_mask([g, g], only_upper_band=True)
    assert np.array_equal(mb.cpu().numpy(), R.generate_pair_mask([g.cpu(), g.cpu()], True).numpy())
    a, at = M.vec_to_matrix_pair(g.reshape(1, -1))
    ra, rat = R.vec_to_matrix_pair(g.cpu())
    assert np.array_equal(a.cpu().numpy(), ra.numpy()) and np.array_equal(at.cpu().numpy(), rat.numpy())


def test_bpr_loss_func_options(dev):
    M = _mod()
    rng = np.random.default_rng(0)
    p, n, w = (rng.normal(size=333).astype(np.float32) for _ in range(3))
    w = np.abs(w)
    for factor in (1.0, 2.5):
        for rm in (True, False):
            pd = torch.from_numpy(p).to(dev).requires_grad_(True)
            nd = torch.from_numpy(n).to(dev).requires_grad_(True)
            out = M.bpr_loss_func(pd, nd, torch.from_numpy(w).to(dev), factor, rm)
            out.backward()
            p64 = torch.from_numpy(p).double().requires_grad_(True)
            n64 = torch.from_numpy(n).double().requires_grad_(True)
            ref = R.bpr_loss_func(p64, n64, torch.from_numpy(w).double(), factor, rm)
            ref.backward()
            assert abs(out.item() - ref.item()) <= RTOL * max(1.0, abs(ref.item()))
            assert np.abs(pd.grad.cpu().numpy() - p64.grad.numpy()).max() <= RTOL * np.abs(p64.grad.numpy()).max()
            assert np.abs(nd.grad.cpu().numpy() - n64.grad.numpy()).max() <= RTOL * np.abs(n64.grad.numpy()).max()


def test_full_size_properties_config3(dev):
    """B=65536 (BASELINE config 3 size): size-independent properties instead of the O(B^2) oracle:
    n_pair == sum_g pos_g*neg_g, sum of gradients == 0 (each pair contributes +-sigma), loss invariant under a
    row permutation, and bitwise run-to-run determinism."""
    M = _mod()
    B = 65536
    rng = np.random.default_rng(3)
    g = rng.integers(0, 1024, B)
    s = rng.normal(size=B).astype(np.float32)
    y = (rng.random(B) < 0.25).astype(np.float32)
    gd, yd = torch.from_numpy(g.astype(np.float32)).to(dev), torch.from_numpy(y).to(dev)
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, n_pair = M.pairwise_loss(sd, yd, gd, return_num_pair=True)
    loss.backward()
    pos_g = np.bincount(g, weights=y, minlength=1024)
    cnt_g = np.bincount(g, minlength=1024)
    assert int(n_pair.item()) == int((pos_g * (cnt_g - pos_g)).sum())
    grad = sd.grad.double().cpu().numpy()
    assert abs(grad.sum()) < 1e-6
    perm = rng.permutation(B)
    loss_p = M.pairwise_loss(torch.from_numpy(s[perm]).to(dev), torch.from_numpy(y[perm]).to(dev),
                             torch.from_numpy(g[perm].astype(np.float32)).to(dev))
    assert abs(loss_p.item() - loss.item()) <= 1e-5 * abs(loss.item())
    loss2 = M.pairwise_loss(sd.detach(), yd, gd)
    assert loss2.item() == loss.item()


def test_pair_indices_bit_exact_vs_c_oracle_large(dev):
    """B = 20000 (beyond what the dense (B,B) oracle can hold comfortably): the HIP pair list must equal the plain-C
    restatement of the reference formulation (oracle/pairs_oracle.c) element for element; fused loss/grad too."""
    import pairs_oracle as C
    M = _mod()
    rng = np.random.default_rng(77)
    B = 20000
    g = rng.integers(0, 300, B).astype(np.float32)
    y = rng.integers(0, 3, B).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    m = rng.random(B) < 0.9
    pos, neg = M.pair_indices(torch.from_numpy(s).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev),
                              mask=torch.from_numpy(m).to(dev))
    cpos, cneg = C.pair_indices(g, y, s, m)
    assert np.array_equal(pos.cpu().numpy(), cpos) and np.array_equal(neg.cpu().numpy(), cneg)
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, n = M.pairwise_loss(sd, torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev), return_num_pair=True,
                              click_occurance_power=-0.5, mask=torch.from_numpy(m).to(dev))
    loss.backward()
    closs, cd, P = C.pairwise_bpr(g, y, s, m, power=-0.5)
    assert int(n.item()) == P == len(cpos)
    assert abs(loss.item() - closs) <= RTOL * abs(closs)
    assert np.abs(sd.grad.cpu().numpy() - cd).max() <= RTOL * np.abs(cd).max()


def _zipf_groups(rng, B, a=1.2, cap=2048):
    """SURVEY 8d, config 2 skewed variant: group sizes ~ Zipf(a) capped at `cap`, rows shuffled."""
    sizes = []
    while sum(sizes) < B:
        sizes.append(int(min(rng.zipf(a), cap, B - sum(sizes))))
    g = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(g)
    return g.astype(np.float32)


@pytest.mark.parametrize('power', [0.0, -0.5])
def test_skewed_group_sizes_vs_c_oracle(dev, power):
    """B = 8192 with Zipf(1.2) group sizes capped at 2048 (a few giant groups beside many singletons): pair list bit-exact
    against the plain-C restatement, loss / gradient within 1e-5."""
    import pairs_oracle as C
    M = _mod()
    rng = np.random.default_rng(2)
    B = 8192
    g = _zipf_groups(rng, B)
    assert np.bincount(g.astype(np.int64)).max() >= 1024          # the case really is skewed
    y = (rng.random(B) < 0.25).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    m = np.ones(B, dtype=bool)
    gd, yd = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev)
    pos, neg = M.pair_indices(torch.from_numpy(s).to(dev), yd, gd)
    cpos, cneg = C.pair_indices(g, y, s, m)
    assert np.array_equal(pos.cpu().numpy(), cpos) and np.array_equal(neg.cpu().numpy(), cneg)
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, n = M.pairwise_loss(sd, yd, gd, return_num_pair=True, click_occurance_power=power)
    loss.backward()
    closs, cd, P = C.pairwise_bpr(g, y, s, m, power=power)
    assert int(n.item()) == P == len(cpos)
    assert abs(loss.item() - closs) <= RTOL * abs(closs)
    assert np.abs(sd.grad.cpu().numpy() - cd).max() <= RTOL * np.abs(cd).max()


def test_empty_batch(dev):
    """B = 0: the reference's dense formulation gives sum over an empty (0,0) mask / (0 + 1e-10) = 0 and no pairs."""
    M = _mod()
    s = torch.zeros(0, device=dev, requires_grad=True)
    y = torch.zeros(0, device=dev)
    g = torch.zeros(0, device=dev)
    loss, n = M.pairwise_loss(s, y, g, return_num_pair=True)
    loss.backward()
    assert loss.item() == 0.0 and n.item() == 0.0 and s.grad.shape == (0,)
    pos, neg = M.pair_indices(s.detach(), y, g)
    assert pos.numel() == 0 and neg.numel() == 0


@pytest.mark.parametrize('wrong', [False, True])
def test_long_and_short_groups_side_by_side(dev, wrong):
    """Groups on both sides of the wave-per-row threshold (512 rows), starting and ending anywhere relative to the 64-row
    blocks of the long-group kernel and the 256-row blocks of the thread-per-row kernels; three label levels, a sample
    mask, occurrence weights.  Loss, pair count and gradient against the plain-C restatement."""
    import pairs_oracle as C
    M = _mod()
    rng = np.random.default_rng(31)
    sizes = [513, 512, 1, 700, 63, 511, 2, 1025, 64, 900, 5, 514]
    g = np.repeat(np.arange(len(sizes)), sizes)
    B = g.size
    rng.shuffle(g)
    g = g.astype(np.float32)
    y = rng.integers(0, 3, B).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    m = rng.random(B) < 0.9
    flags = 3 if wrong else 1
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, n = M.pairwise_loss(sd, torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev), only_use_wrong_order_pair=wrong,
                              return_num_pair=True, click_occurance_power=-0.5, mask=torch.from_numpy(m).to(dev))
    loss.backward()
    closs, cd, P = C.pairwise_bpr(g, y, s, m, flags=flags, power=-0.5)
    assert int(n.item()) == P
    assert abs(loss.item() - closs) <= RTOL * abs(closs)
    assert np.abs(sd.grad.cpu().numpy() - cd).max() <= RTOL * np.abs(cd).max()
    # bitwise reproducible run to run (fixed reduction order in both kernels)
    sd2 = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss2 = M.pairwise_loss(sd2, torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev), only_use_wrong_order_pair=wrong,
                            click_occurance_power=-0.5, mask=torch.from_numpy(m).to(dev))
    loss2.backward()
    assert loss2.item() == loss.item() and torch.equal(sd2.grad, sd.grad)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_groups_beyond_the_lds_staging_size(dev, seed):
    """Groups of more than 2048 rows (the LDS staging size of both walks: the wave-per-row kernel and the thread-per-row
    kernels then read the members from global memory) next to groups of exactly 2048 / 2049 rows and small ones, batch size
    not a multiple of any block size; pair list, count, loss and gradient against the plain-C restatement."""
    import pairs_oracle as C
    M = _mod()
    rng = np.random.default_rng(100 + seed)
    sizes = [5000, 2049, 2048, 37, 1, 3, 600, 2500 + seed, 129]
    g = np.repeat(np.arange(len(sizes)), sizes)
    B = g.size
    rng.shuffle(g)
    g = g.astype(np.float32)
    y = (rng.random(B) < 0.2).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    m = rng.random(B) < 0.95
    gd, yd, md = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(m).to(dev)
    sd = torch.from_numpy(s).to(dev).requires_grad_(True)
    loss, n = M.pairwise_loss(sd, yd, gd, return_num_pair=True, click_occurance_power=-0.5, mask=md)
    loss.backward()
    closs, cd, P = C.pairwise_bpr(g, y, s, m, power=-0.5)
    assert int(n.item()) == P
    assert abs(loss.item() - closs) <= RTOL * abs(closs)
    assert np.abs(sd.grad.cpu().numpy() - cd).max() <= RTOL * np.abs(cd).max()
    if seed == 0:
        pos, neg = M.pair_indices(sd.detach(), yd, gd, mask=md)
        cpos, cneg = C.pair_indices(g, y, s, m)
        assert np.array_equal(pos.cpu().numpy(), cpos) and np.array_equal(neg.cpu().numpy(), cneg)


# ---- the one-launch front end for small batches (recnow_group_pack_small) -----------------------------------------------
def _zipf_groups_cap(rng, B, cap):
    sizes = []
    while sum(sizes) < B:
        sizes.append(int(min(cap, rng.zipf(1.3))))
    g = np.repeat(np.arange(len(sizes)), sizes)[:B]
    return rng.permutation(g).astype(np.float32)


@pytest.mark.parametrize('B,kind,seed', [(8192, 'uniform128', 0), (8192, 'zipf2048', 1), (8192, 'one_group', 2), (8192, 'singletons', 3),
                                          (5000, 'special_ids', 4), (1, 'uniform128', 5), (777, 'two_groups', 6), (8191, 'zipf300', 7)])
@pytest.mark.parametrize('wrong,power,use_mask', [(False, 0.0, False), (True, -0.5, True), (False, 1.0, True)])
def test_small_route_equals_general_route_and_c_oracle(dev, B, kind, seed, wrong, power, use_mask):
    """The one-launch front end (keys + grouping + member packing, B <= 8192, one group tensor) against (a) the general route on
    the same inputs (segments handed in explicitly), (b) the plain-C restatement of the reference formulation (pairs_oracle.c)."""
    import pairs_oracle as PO
    from rec_now_amd.rec_block.pairwise_loss_from_batch import _small_route, group_rows, pairwise_loss_fused
    rng = np.random.default_rng(100 + seed)
    if kind == 'uniform128':
        g = rng.integers(0, 128, B).astype(np.float32)
    elif kind.startswith('zipf'):
        g = _zipf_groups_cap(rng, B, int(kind[4:]))
    elif kind == 'one_group':
        g = np.full(B, 3.0, np.float32)
    elif kind == 'singletons':
        g = rng.permutation(B).astype(np.float32)
    elif kind == 'two_groups':
        g = (rng.random(B) < 0.9).astype(np.float32)
    else:       # NaN / inf ids pair with nobody; -0.0 and +0.0 are one group; negative and fractional ids
        g = rng.choice(np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, -1.5, 2.25, 1e30, -7.0], np.float32), B)
    y = rng.integers(0, 3, B).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    m = (rng.random(B) < 0.85) if use_mask else None
    gd, yd = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev)
    md = None if m is None else torch.from_numpy(m).to(dev)
    assert _small_route(gd) is not None
    outs = []
    for route in ('small', 'general'):
        sd = torch.from_numpy(s).to(dev).requires_grad_(True)
        seg = group_rows(gd) if route == 'general' else None
        loss, n_pair = pairwise_loss_fused(sd, yd, gd, only_use_wrong_order_pair=wrong, click_occurance_power=power, mask=md,
                                           factor=1.3, segments=seg)
        loss.backward()
        outs.append((float(loss), int(n_pair.item()), sd.grad.cpu().numpy()))
    (l1, p1, d1), (l2, p2, d2) = outs
    assert p1 == p2
    assert abs(l1 - l2) <= 2e-6 * max(1.0, abs(l2))
    assert np.abs(d1 - d2).max() <= 2e-6 * max(np.abs(d2).max(), 1e-30)
    flags = 1 | (2 if wrong else 0)
    rl, rd, rp = PO.pairwise_bpr(g, y, s, None if m is None else m.astype(np.uint8), flags, 1.3, power)
    assert p1 == int(np.float32(rp))            # the pair count is returned as a float32 tensor, as the reference returns it
    assert abs(l1 - rl) <= 1e-5 * max(1.0, abs(rl))
    assert np.abs(d1 - rd).max() <= 1e-5 * max(np.abs(rd).max(), 1e-30)


def test_small_route_int_ids_determinism_and_routing(dev):
    from rec_now_amd.rec_block.pairwise_loss_from_batch import _small_route, pairwise_loss
    rng = np.random.default_rng(9)
    B = 6000
    gi = torch.from_numpy(rng.integers(-5, 60, B).astype(np.int32)).to(dev)
    y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)
    s0 = rng.normal(size=B).astype(np.float32)
    res = []
    for g in (gi, gi.to(torch.float32), gi.to(torch.int16)):
        assert _small_route(g) is not None
        sd = torch.from_numpy(s0).to(dev).requires_grad_(True)
        loss = pairwise_loss(sd, y, g)
        loss.backward()
        res.append((loss.detach().clone(), sd.grad.clone()))
    for l, d in res[1:]:
        assert torch.equal(l, res[0][0]) and torch.equal(d, res[0][1])            # same groups whatever the id dtype; bitwise reproducible
    # not this route: two group tensors, 64-bit ids, more than 8192 rows
    assert _small_route([gi, gi]) is None and _small_route(gi.to(torch.int64)) is None
    assert _small_route(torch.zeros(8193, device=dev)) is None


# ---- grouping machinery at mid sizes: the cooperative single-launch path (k_group_mid), several key words ----------------------------
@pytest.mark.parametrize('B,kind', [(30000, 'f32'), (30000, 'pair_f32'), (70000, 'i64'), (9000, 'f64'), (300000, 'f32_wide'),
                                    (524288, 'pair_mixed'), (12345, 'all_equal'),
                                    # round 4: float ids that are all small non-negative integers are sorted by their integer images (fewer digit
                                    # passes), on 512- / 1024- / 2048-key tiles; ids that break the rule (>= 2^24, negative, fractional) as before
                                    (70000, 'int_img'), (262144, 'int_img_big'), (40000, 'int_over'), (40000, 'int_neg'), (40000, 'frac'),
                                    (20000, 'signed_zero_ints'), (8000, 'int_img'), (8000, 'frac'), (8192, 'int_img_big')])
def test_build_segments_invariants_mid_sizes(dev, B, kind):
    """Segments of 8192 < B <= 524288 rows (one cooperative launch) and of multi-word keys: the order is a permutation, rows of a
    segment are ascending (stable sort), rows share a segment iff every key word matches (NaN / inf ids are segments of their own),
    seg_first / n_seg are consistent, and the super segments follow the FIRST group tensor only."""
    from rec_now_amd.rec_block._segments import build_segments
    rng = np.random.default_rng(B % 1000 + len(kind))
    if kind == 'f32':
        gs = [rng.integers(0, 500, B).astype(np.float32)]
        gs[0][::1013] = np.nan
        gs[0][5::2027] = np.inf
    elif kind == 'pair_f32':
        gs = [rng.integers(0, 40, B).astype(np.float32), rng.integers(0, 7, B).astype(np.float32)]
    elif kind == 'i64':
        gs = [rng.integers(-2 ** 40, 2 ** 40, 300)[rng.integers(0, 300, B)].astype(np.int64)]
    elif kind == 'f64':
        gs = [rng.normal(size=200)[rng.integers(0, 200, B)].astype(np.float64)]
    elif kind == 'f32_wide':
        gs = [rng.normal(size=5000).astype(np.float32)[rng.integers(0, 5000, B)]]
    elif kind == 'pair_mixed':
        gs = [rng.integers(0, 3000, B).astype(np.int32), rng.integers(0, 3, B).astype(np.float32)]
    elif kind == 'int_img':
        gs = [rng.integers(0, 4096, B).astype(np.float32)]
    elif kind == 'int_img_big':
        gs = [rng.integers(0, 2 ** 24, 3000)[rng.integers(0, 3000, B)].astype(np.float32)]
        gs[0][:4] = [0.0, 16777215.0, 1.0, 16777215.0]
    elif kind == 'int_over':
        gs = [rng.integers(0, 300, B).astype(np.float32)]
        gs[0][::97] = 16777216.0 + 2.0 * rng.integers(0, 5, len(gs[0][::97]))
    elif kind == 'int_neg':
        gs = [rng.integers(-50, 50, B).astype(np.float32)]
    elif kind == 'frac':
        gs = [rng.integers(0, 300, B).astype(np.float32) + 0.5]
    elif kind == 'signed_zero_ints':
        gs = [rng.integers(0, 10, B).astype(np.float32)]
        gs[0][gs[0] == 0][::2] = -0.0
        gs[0][::31] = -0.0
    else:
        gs = [np.full(B, -0.0, np.float32)]
        gs[0][::2] = 0.0                                            # -0.0 and +0.0 are one group
    seg = build_segments([torch.from_numpy(g).to(dev) for g in gs])
    order, seg_id, super_id = (t.cpu().numpy()[:B] for t in (seg.order, seg.seg_id, seg.super_id))
    n_seg, n_super = (int(v) for v in seg.n_seg.cpu().numpy())
    seg_first = seg.seg_first.cpu().numpy()[:n_seg + 1]
    assert n_seg >= 1 and np.array_equal(np.sort(order), np.arange(B))
    assert seg_first[0] == 0 and seg_first[-1] == B and np.all(np.diff(seg_first) > 0)
    assert np.array_equal(seg_id, np.repeat(np.arange(n_seg), np.diff(seg_first)))
    same_seg = seg_id[1:] == seg_id[:-1]
    assert np.all(order[1:][same_seg] > order[:-1][same_seg])                       # stable: ascending rows inside a segment

    def canon(g):
        g = g[order]
        solo = ~np.isfinite(g) if g.dtype.kind == 'f' else np.zeros(B, bool)
        return np.where(g == 0, 0, g) if g.dtype.kind == 'f' else g, solo

    keys = [canon(g) for g in gs]
    solo = np.zeros(B, bool)
    for _, so in keys:
        solo |= so
    eq_all = np.ones(B - 1, bool)
    for k, _ in keys:
        eq_all &= k[1:] == k[:-1]
    eq_all &= ~(solo[1:] | solo[:-1])
    assert np.array_equal(same_seg, eq_all)                                           # neighbours share a segment iff all words match
    # globally: the number of segments = distinct composite keys among the finite rows + one per solo row
    fin = ~solo
    comp = np.stack([k[fin].astype(np.float64) if k.dtype.kind == 'f' else k[fin] for k, _ in keys], 1)
    assert n_seg == len(np.unique(comp, axis=0)) + int(solo.sum())
    eq_first = (keys[0][0][1:] == keys[0][0][:-1]) & ~(solo[1:] | solo[:-1])
    assert np.array_equal(super_id[1:] == super_id[:-1], eq_first) and n_super == super_id[-1] + 1


@pytest.mark.parametrize('reduce_mean', [True, False])
def test_one_pass_route_equals_two_pass_route(dev, reduce_mean):
    """click_occurance_power == 0 takes the one-walk route (recnow_pair_bpr_onepass + recnow_pair_scale_grad); a power of 1e-30
    gives weights cnt ** 1e-30 = 1 to fp32 precision but takes the count-then-loss route: same loss, same pair count, same gradient,
    also under a non-unit incoming gradient (the 1 / (P + 1e-10) of the mean is applied where it is multiplied in)."""
    from rec_now_amd import _lib
    M = _mod()
    rng = np.random.default_rng(12)
    B = 20000
    s = rng.normal(size=B).astype(np.float32)
    y = rng.integers(0, 3, B).astype(np.float32)
    g = np.concatenate([rng.integers(0, 300, B - 3000), np.full(3000, 999)]).astype(np.int64)       # one 3000-row group: the wave-per-row walks
    rng.shuffle(g)
    mask = rng.random(B) < 0.9
    res = []
    for power in (0.0, 1e-30):
        sd = torch.from_numpy(s).to(dev).requires_grad_(True)
        loss, n_pair = M.pairwise_loss_fused(sd, torch.from_numpy(y).to(dev), torch.from_numpy(g).to(dev), click_occurance_power=power,
                                             mask=torch.from_numpy(mask).to(dev), reduce_mean=reduce_mean)
        (loss * 2.5).backward()
        res.append((float(loss), float(n_pair), sd.grad.detach().cpu().numpy()))
    (l0, p0, g0), (l1, p1, g1) = res
    assert p0 == p1 and p0 > 0
    assert abs(l0 - l1) <= 2e-6 * abs(l1)
    assert np.abs(g0 - g1).max() <= 2e-6 * np.abs(g1).max()
    # the C entry point refuses a pair set without the label / wrong-order predicate, as recnow_pair_bpr_fwdbwd does
    z = torch.zeros(8, device=dev)
    zi = torch.zeros(9, dtype=torch.int32, device=dev)
    ws = _lib.workspace(_lib.load().recnow_pairwise_workspace_bytes(8), dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = _lib.load().recnow_pair_bpr_onepass(_lib.ptr(z), _lib.ptr(z), None, _lib.ptr(zi), _lib.ptr(zi), _lib.ptr(zi), 8, 0, 1.0, 1,
                                             _lib.ptr(z), _lib.ptr(z), _lib.ptr(n), _lib.ptr(ws), ws.numel(), _lib.stream())
    assert rc != 0


def test_grouping_timeout_poisons_the_one_call_losses(dev):
    """when the cooperative grouping launch times out at a grid barrier it leaves the identity grouping and n_seg = -1; the
    one-call consumers (pairwise_loss, listwise loss, the whole-step entry) must not turn that into `loss 0, zero gradients, RECNOW_OK`.
    RECNOW_DEBUG_GROUP_TIMEOUT=1 makes every barrier of k_group_mid report the time-out (own process: the switch is read once)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss
from rec_now_amd.rec_block.listwise_loss_from_batch import listwise_loss_from_batch
from rec_now_amd.rec_block._segments import build_segments
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
B = 20000                                     # above the one-workgroup route (8192), inside the cooperative one
g = torch.from_numpy(rng.integers(0, 300, B).astype(np.float32)).to(dev)
y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)
s = torch.from_numpy(rng.normal(size=B).astype(np.float32)).to(dev).requires_grad_(True)
try:
    build_segments(g).num_segments()
    raise SystemExit('num_segments() did not raise')
except RuntimeError as e:
    assert 'timed out' in str(e)
loss = pairwise_loss(s, y, g)
loss.backward()
assert torch.isnan(loss).item() and torch.isnan(s.grad).all().item(), (loss, s.grad[:4])
s2 = s.detach().clone().requires_grad_(True)
lw = listwise_loss_from_batch(g, y, s2)
lw.backward()
assert torch.isnan(lw).item() and torch.isnan(s2.grad).all().item(), (lw, s2.grad[:4])
print('poisoned ok')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=root,
                         env=dict(os.environ, RECNOW_DEBUG_GROUP_TIMEOUT='1', PYTHONPATH=root))
    assert out.returncode == 0 and 'poisoned ok' in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_cooperative_grouping_beside_a_cu_filling_kernel(dev):
    """the single-launch grouping (k_group_mid: a plain launch whose grid barrier counts on its workgroups being co-resident, gated
    on an occupancy query) started while ANOTHER stream keeps every CU busy with long fp32 products -- the situation of a grouping launched beside the
    step's GEMMs or RCCL's persistent kernels.  Its workgroups then become resident as slots free up; the bounded barrier must not time out
    (n_seg >= 0) and the segments must equal those of a quiet run.  65 536 rows, ~1024 groups."""
    from rec_now_amd.rec_block.pairwise_loss_from_batch import group_rows
    rng = np.random.default_rng(123)
    B = 65536
    g = torch.from_numpy(rng.integers(0, 1024, B).astype(np.float32)).to(dev)
    quiet = group_rows(g)
    torch.cuda.synchronize()
    n_quiet = quiet.num_segments()
    a = torch.randn(8192, 8192, device=dev)
    busy, side = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(busy):
        for _ in range(6):                 # ~50 ms of CU-filling work
            a = (a @ a) * 1e-4
    with torch.cuda.stream(side):
        under = [group_rows(g) for _ in range(4)]
    torch.cuda.synchronize()
    for s in under:
        assert s.num_segments() == n_quiet >= 1000          # raises on a barrier time-out (n_seg = -1)
        assert torch.equal(s.order, quiet.order) and torch.equal(s.seg_id, quiet.seg_id) and torch.equal(s.seg_first[:n_quiet + 1], quiet.seg_first[:n_quiet + 1])
