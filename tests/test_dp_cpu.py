"""CPU, world_size 2 over gloo: the data-parallel combination rules of rec_now_amd/dp.py.
With every group living on one rank, the 2-rank global loss / gradient scaling / weight-gradient sum must equal the
single-process result on the concatenated batch.  (The kernels themselves need a GPU; here the per-rank loss SUMS are
produced by the oracle so that only the distributed logic is under test.)"""
import os
import socket

import numpy as np
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
