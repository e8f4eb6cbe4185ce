"""CPU, world_size 2 over gloo: the data-parallel combination rules of rec_now_amd/dp.py.
With every group living on one rank, the 2-rank global loss / gradient scaling / weight-gradient sum must equal the
single-process result on the concatenated batch.  (The kernels themselves need a GPU; here the per-rank loss SUMS are
produced by the oracle so that only the distributed logic is under test.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dense_ref as R


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(seed=0, B=400, G=23):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, G, B)
    s = rng.normal(size=B).astype(np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    w = rng.normal(size=(4,)).astype(np.float32)
    return g, s, y, w


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rec_now_amd import dp
    g, s, y, w = _make_batch()
    owner = dp.shard_rows_by_group(g, world).numpy()
    mine = owner == rank
    # a tiny "model": score = s * w0 + w1 (replicated weights), loss on the local shard
    wt = torch.from_numpy(w.copy()).requires_grad_(True)
    sc = torch.from_numpy(s[mine]) * wt[0] + wt[1]
    gl = torch.from_numpy(g[mine].astype(np.float32))
    yl = torch.from_numpy(y[mine])
    f = lambda p, n, wgt: R.bpr_loss_func(p, n, wgt, 1.0, reduce_mean=False)      # noqa: E731  local SUM
    local_sum, n_pair = R.pairwise_loss(sc, yl, gl, f, return_num_pair=True)
    loss_bw, loss_val, p_glob = dp.global_pairwise_loss(local_sum, torch.tensor(n_pair))
    loss_bw.backward()
    red = dp.GradientAllReducer([wt])
    red.all_reduce()
    # the one-collective form: backward on the unnormalised local sum, statistics ride along with the gradients
    wt2 = torch.from_numpy(w.copy()).requires_grad_(True)
    sc2 = torch.from_numpy(s[mine]) * wt2[0] + wt2[1]
    local_sum2, n_pair2 = R.pairwise_loss(sc2, yl, gl, f, return_num_pair=True)
    local_sum2.backward()
    loss_val2, p_glob2 = dp.GradientAllReducer([wt2]).all_reduce_with_loss(local_sum2, torch.tensor(n_pair2))
    assert abs(float(loss_val2) - float(loss_val)) <= 1e-6 * max(1.0, abs(float(loss_val))) and float(p_glob2) == float(p_glob)
    assert np.abs(wt2.grad.numpy() - wt.grad.numpy()).max() <= 1e-6 * max(1.0, np.abs(wt.grad.numpy()).max())
    out[rank] = (float(loss_val), float(p_glob), wt.grad.numpy().copy(), int(mine.sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_loss_and_grads_equal_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g, s, y, w = _make_batch()
    wt = torch.from_numpy(w.copy()).requires_grad_(True)
    sc = torch.from_numpy(s) * wt[0] + wt[1]
    loss, n_pair = R.pairwise_loss(sc, torch.from_numpy(y), torch.from_numpy(g.astype(np.float32)), return_num_pair=True)
    loss.backward()
    assert out[0][3] + out[1][3] == len(g) and out[0][3] > 0 and out[1][3] > 0
    for r in range(world):
        lv, pg, grad, _ = out[r]
        assert pg == n_pair
        assert abs(lv - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
        assert np.abs(grad - wt.grad.numpy()).max() <= 1e-6 * max(1.0, np.abs(wt.grad.numpy()).max())


def test_shard_rows_by_group_keeps_groups_whole():
    from rec_now_amd import dp
    g = np.random.default_rng(1).integers(0, 1000, 20000)
    owner = dp.shard_rows_by_group(g, 8).numpy()
    for gid in np.unique(g)[:200]:
        assert len(np.unique(owner[g == gid])) == 1
    counts = np.bincount(owner, minlength=8)
    assert counts.min() > 0.5 * counts.mean()


def test_single_process_passthrough():
    from rec_now_amd import dp
    s = torch.tensor(6.0, requires_grad=True)
    lb, lv, p = dp.global_pairwise_loss(s, torch.tensor(3.0))
    assert abs(float(lv) - 2.0) < 1e-6 and float(p) == 3.0
    lb.backward()
    assert abs(float(s.grad) - 1.0 / 3.0) < 1e-6
    lb, lv, n = dp.global_listwise_loss(torch.tensor(0.0), torch.tensor(0.0))
    assert float(lv) == 0.0
    w = torch.tensor([2.0], requires_grad=True)
    ls = (w * 3.0).sum()
    ls.backward()
    lv, p = dp.GradientAllReducer([w]).all_reduce_with_loss(ls, torch.tensor(3.0))
    assert abs(float(lv) - 2.0) < 1e-6 and float(p) == 3.0 and abs(float(w.grad) - 1.0) < 1e-6


# ---- world size 4: one-collective form, listwise combination, uneven shards, an empty rank ---------------------------------
def _owner_uneven(g, world, empty_rank=None):
    """Whole groups per rank, deliberately uneven: rank 0 owns half of the groups; `empty_rank` owns none."""
    ranks = [r for r in range(world) if r != empty_rank]
    uniq = np.unique(g)
    table = {}
    for i, gid in enumerate(uniq):
        table[gid] = ranks[0] if i % 2 == 0 else ranks[1 + (i // 2) % (len(ranks) - 1)]
    return np.array([table[v] for v in g])


def _listwise_sums(g, y, s):
    """(sum over valid lists of the list loss, number of valid lists) from the oracle, attached to s's graph."""
    _, rl, rz = R.to_listwise_sample(g, y, s)
    if rl.shape[0] == 0:
        return s.sum() * 0.0, 0
    per_list = R.listwise_loss_via_softmax_cross_entropy_with_logits(rl, rz, do_reduce=False)
    return per_list.sum(), int(per_list.shape[0])


def _worker4(rank, world, port, out, empty_rank):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rec_now_amd import dp
    g, s, y, w = _make_batch(seed=7, B=900, G=41)
    mine = _owner_uneven(g, world, empty_rank) == rank
    gl, yl = torch.from_numpy(g[mine].astype(np.float32)), torch.from_numpy(y[mine])
    # two parameters in two buckets (bucket_bytes = 8 -> one parameter per bucket): the statistics ride in the LAST bucket
    wa = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    wb = torch.from_numpy(w[2:3].copy()).requires_grad_(True)
    sc = torch.from_numpy(s[mine]) * wa[0] + wa[1] + wb[0] * torch.from_numpy(s[mine]) ** 2
    f = lambda p, n, wgt: R.bpr_loss_func(p, n, wgt, 1.0, reduce_mean=False)      # noqa: E731
    if mine.sum() > 0:
        local_sum, n_pair = R.pairwise_loss(sc, yl, gl, f, return_num_pair=True)
    else:
        local_sum, n_pair = sc.sum() * 0.0, 0.0                                     # a rank without rows still joins the collective
    local_sum.backward()
    red = dp.GradientAllReducer([wa, wb], bucket_bytes=8)
    assert len(red.buckets) == 2
    loss_val, p_glob = red.all_reduce_with_loss(local_sum, torch.tensor(float(n_pair)))
    # listwise: per-rank sums of list losses + valid-list counts -> global mean over all valid lists
    wl = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    sl = torch.from_numpy(s[mine]) * wl[0] + wl[1]
    lsum, nv = _listwise_sums(gl, yl, sl)
    lw_bw, lw_val, nv_glob = dp.global_listwise_loss(lsum, torch.tensor(float(nv)))
    lw_bw.backward()
    red2 = dp.GradientAllReducer([wl])
    red2.all_reduce()
    # layer-wise (overlapped) form: two stages, the statistics ride in the first bucket, every bucket scaled after its collective
    wc = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    wd = torch.from_numpy(w[2:3].copy()).requires_grad_(True)
    sc3 = torch.from_numpy(s[mine]) * wc[0] + wc[1] + wd[0] * torch.from_numpy(s[mine]) ** 2
    if mine.sum() > 0:
        local_sum3, n_pair3 = R.pairwise_loss(sc3, yl, gl, f, return_num_pair=True)
    else:
        local_sum3, n_pair3 = sc3.sum() * 0.0, 0.0
    local_sum3.backward()
    lw_red = dp.LayerwiseReducer([[wd], [wc]], [None, None], 'cpu')
    loss_val3, p_glob3 = lw_red.reduce(local_sum3, torch.tensor(float(n_pair3)))
    assert abs(float(loss_val3) - float(loss_val)) <= 2e-6 * max(1.0, abs(float(loss_val))) and float(p_glob3) == float(p_glob)
    assert np.abs(wc.grad.numpy() - wa.grad.numpy()).max() <= 2e-6 * max(1.0, np.abs(wa.grad.numpy()).max())
    assert np.abs(wd.grad.numpy() - wb.grad.numpy()).max() <= 2e-6 * max(1.0, np.abs(wb.grad.numpy()).max())
    out[rank] = (float(loss_val), float(p_glob), wa.grad.numpy().copy(), wb.grad.numpy().copy(), int(mine.sum()),
                 float(lw_val), float(nv_glob), wl.grad.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _run4(empty_rank):
    world = 4
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker4, args=(world, port, out, empty_rank), nprocs=world, join=True)
    g, s, y, w = _make_batch(seed=7, B=900, G=41)
    wa = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    wb = torch.from_numpy(w[2:3].copy()).requires_grad_(True)
    sc = torch.from_numpy(s) * wa[0] + wa[1] + wb[0] * torch.from_numpy(s) ** 2
    gt, yt = torch.from_numpy(g.astype(np.float32)), torch.from_numpy(y)
    loss, n_pair = R.pairwise_loss(sc, yt, gt, return_num_pair=True)
    loss.backward()
    wl = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    _, rl, rz = R.to_listwise_sample(gt, yt, torch.from_numpy(s) * wl[0] + wl[1])
    lw = R.listwise_loss_via_softmax_cross_entropy_with_logits(rl, rz)
    lw.backward()
    sizes = [out[r][4] for r in range(world)]
    assert sum(sizes) == len(g) and len(set(sizes)) > 1                       # uneven shards
    if empty_rank is not None:
        assert sizes[empty_rank] == 0
    rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())      # noqa: E731
    for r in range(world):
        lv, pg, ga, gb, _, lwv, nvg, gl = out[r]
        assert pg == n_pair
        assert rel(lv, float(loss)) <= 2e-6
        assert rel(ga, wa.grad.numpy()) <= 2e-6 and rel(gb, wb.grad.numpy()) <= 2e-6
        assert nvg == rl.shape[0]
        assert rel(lwv, float(lw)) <= 2e-6
        assert rel(gl, wl.grad.numpy()) <= 2e-6


def test_four_ranks_uneven_shards():
    _run4(None)


def test_four_ranks_with_an_empty_rank():
    _run4(2)


# ---- C3 exact variant: arbitrarily sharded rows, one all-gather of (score, label, group) triples -----------------------------
def _worker_gather(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rec_now_amd import dp
    g, s, y, w = _make_batch(seed=11, B=700, G=17)
    owner = np.random.default_rng(5).integers(0, world + 1, len(g)) % world      # rows dealt WITHOUT regard to their group, unevenly
    owner[owner == world - 1] = 0 if world > 2 else owner[owner == world - 1]
    mine = owner == rank
    wt = torch.from_numpy(w.copy()).requires_grad_(True)
    sc = torch.from_numpy(s[mine]) * wt[0] + wt[1] + wt[2] * torch.from_numpy(s[mine]) ** 2
    loss = dp.gathered_pairwise_loss(sc, torch.from_numpy(y[mine]), torch.from_numpy(g[mine].astype(np.float32)), loss_fn=R.pairwise_loss)
    loss.backward()
    dp.GradientAllReducer([wt]).all_reduce()
    out[rank] = (float(loss), wt.grad.numpy().copy(), int(mine.sum()), len(np.intersect1d(g[mine], g[~mine])))
    dist.barrier()
    dist.destroy_process_group()


def test_arbitrarily_sharded_rows_equal_single_process_through_all_gather():
    """dp.gathered_pairwise_loss (variant C3): groups are SPLIT across the ranks, yet the loss and the weight gradients
    equal the single-process ones on the whole batch (rec_block/pairwise_loss_from_batch.py:254-274 on the concatenation)."""
    for world in (2, 3):
        port = _free_port()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker_gather, args=(world, port, out), nprocs=world, join=True)
        g, s, y, w = _make_batch(seed=11, B=700, G=17)
        wt = torch.from_numpy(w.copy()).requires_grad_(True)
        sc = torch.from_numpy(s) * wt[0] + wt[1] + wt[2] * torch.from_numpy(s) ** 2
        loss = R.pairwise_loss(sc, torch.from_numpy(y), torch.from_numpy(g.astype(np.float32)))
        loss.backward()
        assert sum(out[r][2] for r in range(world)) == len(g)
        assert out[0][3] > 5                                      # groups really are shared between ranks
        for r in range(world):
            assert abs(out[r][0] - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
            assert np.abs(out[r][1] - wt.grad.numpy()).max() <= 2e-6 * max(1.0, np.abs(wt.grad.numpy()).max())


def _int_id_cases():
    """Group ids that a float block would merge: int32 above 2^24 (float32 rounds 2^24 + 1 to 2^24), int64 above 2^53 (float64), and the
    reference's LIST of group tensors (pairs must agree on every one)."""
    g, s, y, w = _make_batch(seed=17, B=500, G=23)
    gi = g.astype(np.int64)
    return s, y, w, {'int32': torch.from_numpy(((1 << 24) + gi).astype(np.int32)),
                     'int64': torch.from_numpy((1 << 53) + gi),
                     'list': [torch.from_numpy((gi // 3).astype(np.float32)), torch.from_numpy(((1 << 24) + gi % 3).astype(np.int32))]}


def _take(groups, idx):
    return [t[idx] for t in groups] if isinstance(groups, list) else groups[idx]


def _worker_gather_ids(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rec_now_amd import dp
    s, y, w, cases = _int_id_cases()
    mine = torch.from_numpy(np.random.default_rng(9).integers(0, world, len(s)) == rank)
    res = {}
    for name, groups in cases.items():
        wt = torch.from_numpy(w.copy()).requires_grad_(True)
        sc = torch.from_numpy(s)[mine] * wt[0] + wt[1]
        loss = dp.gathered_pairwise_loss(sc, torch.from_numpy(y)[mine], _take(groups, mine), loss_fn=R.pairwise_loss)
        loss.backward()
        dp.GradientAllReducer([wt]).all_reduce()
        res[name] = (float(loss), wt.grad.numpy().copy())
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_loss_keeps_integer_ids_exact_and_accepts_group_lists():
    """ids travel in their native dtype, so int32 ids above 2^24 and int64 ids above 2^53 stay distinct groups after the
    all-gather; a list of group tensors is gathered tensor by tensor."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_gather_ids, args=(world, port, out), nprocs=world, join=True)
    s, y, w, cases = _int_id_cases()
    merged = {}
    for name, groups in cases.items():
        wt = torch.from_numpy(w.copy()).requires_grad_(True)
        sc = torch.from_numpy(s) * wt[0] + wt[1]
        loss, n_pair = R.pairwise_loss(sc, torch.from_numpy(y), groups, return_num_pair=True)
        loss.backward()
        for r in range(world):
            assert abs(out[r][name][0] - float(loss)) <= 1e-6 * max(1.0, abs(float(loss))), name
            assert np.abs(out[r][name][1] - wt.grad.numpy()).max() <= 2e-6 * max(1.0, np.abs(wt.grad.numpy()).max()), name
        if name != 'list':      # the same ids through a float block of the old width: groups merge, another pair count -- the bug this guards against
            lossy = groups.to(torch.float32 if name == 'int32' else torch.float64)
            merged[name] = int(R.pairwise_loss(sc.detach(), torch.from_numpy(y), lossy, return_num_pair=True)[1]) != int(n_pair)
    assert merged['int32'] and merged['int64']


# ---- in-place protocol of step.DCNMixPairwiseStep: gradients and statistics produced inside the buckets ---------------------
def _worker_inplace(rank, world, port, out, one_collective=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rec_now_amd import dp
    g, s, y, w = _make_batch(seed=13, B=600, G=29)
    mine = dp.shard_rows_by_group(g, world).numpy() == rank
    wa = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    wb = torch.from_numpy(w[2:3].copy()).requires_grad_(True)
    sc = torch.from_numpy(s[mine]) * wa[0] + wa[1] + wb[0] * torch.from_numpy(s[mine]) ** 2
    f = lambda p, n, wgt: R.bpr_loss_func(p, n, wgt, 1.0, reduce_mean=False)      # noqa: E731
    local_sum, n_pair = R.pairwise_loss(sc, torch.from_numpy(y[mine]), torch.from_numpy(g[mine].astype(np.float32)), f, return_num_pair=True)
    ga, gb = torch.autograd.grad(local_sum, [wa, wb])
    red = dp.LayerwiseReducer([[wb], [wa]], [None, None], 'cpu', one_collective=one_collective)
    # what the step's kernels do on the GPU: gradients written into the bucket slices, {loss sum, pair count} into the slot
    red.buffer_of(wb).copy_(gb.reshape(-1))
    red.buffer_of(wa).copy_(ga.reshape(-1))
    red.stats_slot().copy_(torch.tensor([float(local_sum), float(n_pair)]))
    wa.grad, wb.grad = red.buffer_of(wa).view(wa.shape), red.buffer_of(wb).view(wb.shape)
    red.stage_done(0)
    red.stage_done(1)
    loss_val, p_glob = red.reduce_in_place()
    out[rank] = (float(loss_val), float(p_glob), wa.grad.numpy().copy(), wb.grad.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('one_collective', [False, True])
def test_in_place_reducer_protocol_equals_single_process(one_collective):
    """one_collective: ONE all-reduce over all stages behind the last one (RECNOW_DP_ONE_BUCKET=1) instead of one per stage."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_inplace, args=(world, port, out, one_collective), nprocs=world, join=True)
    g, s, y, w = _make_batch(seed=13, B=600, G=29)
    wa = torch.from_numpy(w[:2].copy()).requires_grad_(True)
    wb = torch.from_numpy(w[2:3].copy()).requires_grad_(True)
    sc = torch.from_numpy(s) * wa[0] + wa[1] + wb[0] * torch.from_numpy(s) ** 2
    loss, n_pair = R.pairwise_loss(sc, torch.from_numpy(y), torch.from_numpy(g.astype(np.float32)), return_num_pair=True)
    loss.backward()
    for r in range(world):
        lv, pg, ga, gb = out[r]
        assert pg == n_pair and abs(lv - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
        assert np.abs(ga - wa.grad.numpy()).max() <= 2e-6 * max(1.0, np.abs(wa.grad.numpy()).max())
        assert np.abs(gb - wb.grad.numpy()).max() <= 2e-6 * max(1.0, np.abs(wb.grad.numpy()).max())


def _mlp_params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(6, 16, generator=g).mul_(0.3).requires_grad_(True), torch.zeros(16).requires_grad_(True),
            torch.randn(16, 1, generator=g).mul_(0.3).requires_grad_(True), torch.zeros(1).requires_grad_(True),
            torch.randn(3, 3, generator=g).requires_grad_(True)]          # the last one never gets a gradient: its bucket is reduced as zeros


def _mlp_scores(params, x):
    return (torch.tanh(x @ params[0] + params[1]) @ params[2] + params[3]).reshape(-1)


def _overlap_worker(rank, world, port, out, kind):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rec_now_amd import dp
    rng = np.random.default_rng(7)
    B, G = 600, 31
    g = rng.integers(0, G, B)
    x = rng.normal(size=(B, 6)).astype(np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    mine = dp.shard_rows_by_group(g, world).numpy() == rank
    params = _mlp_params(5)
    red = dp.OverlappedGradientReducer(params, bucket_bytes=100, denom='eps' if kind == 'pairwise' else 'max1')      # several small buckets
    assert len(red.buckets) >= 3
    for step in range(2):                 # twice: the buckets and hooks are reused, the second step must not see the first one's gradients
        for p in params:
            p.grad = None
        sc = _mlp_scores(params, torch.from_numpy(x[mine]))
        gl, yl = torch.from_numpy(g[mine].astype(np.float32)), torch.from_numpy(y[mine])
        if kind == 'pairwise':
            f = lambda p, n, wgt: R.bpr_loss_func(p, n, wgt, 1.0, reduce_mean=False)      # noqa: E731
            local_sum, cnt = R.pairwise_loss(sc, yl, gl, f, return_num_pair=True)
            cnt = torch.tensor(float(cnt))
        else:
            _, lab, lg = R.to_listwise_sample(gl, yl.double(), sc.double())
            per_list = R.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg, do_reduce=False)
            local_sum, cnt = per_list.sum().float(), torch.tensor(float(lab.shape[0]))
        red.prepare(local_sum, cnt)
        local_sum.backward()
        loss, total = red.finish()
    out[rank] = (float(loss), float(total), [p.grad.numpy().copy() for p in params])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['pairwise', 'listwise'])
def test_overlapped_reducer_two_ranks_equal_single_process(kind):
    """dp.OverlappedGradientReducer (round 6: the reducer of the configs[3] / configs[4] model steps): buckets in reverse registration order, every
    bucket all-reduced from the post-accumulate-grad hook that completes it, the loss statistics in the first bucket, a parameter without a
    gradient reduced as zeros -- two gloo ranks with whole groups per rank against the single-process loss and gradients, pairwise (1 / (P + 1e-10))
    and listwise (1 / max(lists, 1)) normalisation."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, port, out, kind), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    B, G = 600, 31
    g = rng.integers(0, G, B)
    x = rng.normal(size=(B, 6)).astype(np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    params = _mlp_params(5)
    sc = _mlp_scores(params, torch.from_numpy(x))
    gt, yt = torch.from_numpy(g.astype(np.float32)), torch.from_numpy(y)
    if kind == 'pairwise':
        loss, cnt = R.pairwise_loss(sc, yt, gt, return_num_pair=True)
    else:
        _, lab, lg = R.to_listwise_sample(gt, yt.double(), sc.double())
        loss, cnt = R.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg).float(), lab.shape[0]
    loss.backward()
    for r in range(world):
        lv, total, grads = out[r]
        assert total == float(cnt)
        assert abs(lv - float(loss)) <= 2e-6 * max(1.0, abs(float(loss)))
        for gp, p in zip(grads, params):
            ref = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
            assert np.abs(gp - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
