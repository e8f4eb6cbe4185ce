"""GPU parity of the model-level fused node `dcn_mix_score` (rec_now_amd/fused.py: DCNMixLayer + MultiDenseLayer(1,1) head as one
autograd node) against (a) the fp64 oracle (reference /root/reference/rec_now/layers/dcn_mix_layer.py:114-151 ->
multi_dense_layer.py:80-94) and (b) the unfused composition of the two drop-in layers, forward and every gradient.  The full-size
oracle comparison lives in test_northstar_gpu.py (route 'fused')."""
import numpy as np
import pytest
import torch

import dense_ref as R
from _chunked_oracle import close

pytestmark = pytest.mark.gpu


def _build(dev, B, D, S, N, L, seed, ai='tanh', ao='tanh'):
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.6, (B, D)).astype(np.float32)
    cross, head = DCNMixLayer(S, num_layer=L, num_expert=N, activation_inner=ai, activation_outer=ao), MultiDenseLayer(1, 1)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    head(cross(xd[:256]))
    w = {k: rng.uniform(-1, 1, tuple(v.shape)).astype(np.float32) * (0.3 if 'bias' in k else 1.5 / np.sqrt(v.shape[-2]))
         for k, v in cross.named_weights().items()}
    cross.set_weights_by_name(w)
    hk = rng.uniform(-1, 1, (1, D, 1)).astype(np.float32)
    hb = np.array([[[0.37]]], np.float32)
    head.set_weights_by_name({'kernel': hk, 'bias': hb})
    return x, xd, cross, head, w, hk, hb


def _oracle(x, w, hk, hb, L, gs, ai='tanh', ao='tanh'):
    w64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    hk64, hb64 = torch.from_numpy(hk).double().requires_grad_(True), torch.from_numpy(hb).double().requires_grad_(True)
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    pick = lambda fmt: [w64[fmt % l] for l in range(L)]     # noqa: E731
    y = R.dcn_mix_layer(x64, pick('origin_to_sub_kernels_of_layer%d'), pick('sub_to_sub_kernels_of_layer%d'),
                        pick('sub_to_origin_kernels_of_layer%d'), pick('bias_of_layer%d'), pick('gate_of_layer%d/kernel'), ai, ao)
    s = R.multi_dense_layer(y, hk64, hb64).reshape(-1)
    s.backward(torch.from_numpy(gs).double())
    return s, x64, w64, hk64, hb64


@pytest.mark.parametrize('B,D,S,N,L,ai,ao', [(512, 256, 64, 2, 2, 'tanh', 'tanh'), (768, 128, 64, 2, 3, 'tanh', 'tanh'),
                                              (256, 384, 64, 2, 1, 'tanh', 'tanh'), (1024, 256, 32, 4, 2, 'relu', 'sigmoid'),
                                              (512, 1024, 64, 2, 3, None, 'tanh')])
def test_fused_score_vs_oracle_and_unfused(dev, B, D, S, N, L, ai, ao):
    from rec_now_amd.fused import dcn_mix_score, fused_route_available
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, S, N, L, B + D + L, ai, ao)
    gs = np.random.default_rng(1).normal(size=B).astype(np.float32)
    assert fused_route_available(cross, head, xd)
    s = dcn_mix_score(cross, head, xd)
    s.backward(torch.from_numpy(gs).to(dev))
    rs, x64, w64, hk64, hb64 = _oracle(x, w, hk, hb, L, gs, ai, ao)
    close(s, rs, what='scores')
    close(xd.grad, x64.grad, what='dx')
    for k, p in cross.named_weights().items():
        close(p.grad, w64[k].grad, what=k)
    close(head.kernel.grad, hk64.grad, what='head kernel')
    close(head.bias.grad, hb64.grad, what='head bias', scale=np.abs(gs).sum())
    fused = {k: p.grad.clone() for k, p in cross.named_weights().items()}
    fused['hk'], fused['hb'], fused['dx'] = head.kernel.grad.clone(), head.bias.grad.clone(), xd.grad.clone()
    for p in list(cross.parameters()) + list(head.parameters()):
        p.grad = None
    xd.grad = None
    s2 = head(cross(xd)).reshape(-1)
    s2.backward(torch.from_numpy(gs).to(dev))
    close(s, s2, rtol=2e-6, what='scores fused vs layers')
    close(fused['dx'], xd.grad, rtol=3e-6, what='dx fused vs layers')
    for k, p in cross.named_weights().items():
        close(fused[k], p.grad, rtol=3e-6, what=k + ' fused vs layers')
    close(fused['hk'], head.kernel.grad, rtol=3e-6, what='head kernel fused vs layers')


def test_fused_score_without_input_gradient_and_events(dev):
    """x as data (no dx product, nothing kept for it) and the per-layer events: a side stream that waits for event l sees the
    final gradients of layer l."""
    from rec_now_amd.fused import GpuEvent, dcn_mix_score
    B, D, S, N, L = 1024, 256, 64, 2, 3
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, S, N, L, 5)
    gs = np.random.default_rng(2).normal(size=B).astype(np.float32)
    xn = xd.detach()
    events = [GpuEvent() for _ in range(L)]
    s = dcn_mix_score(cross, head, xn, layer_events=events)
    s.backward(torch.from_numpy(gs).to(dev))
    side = torch.cuda.Stream(device=dev)
    snap = {}
    with torch.cuda.stream(side):
        for l in range(L - 1, -1, -1):
            events[l].wait(side)
            snap[l] = cross.named_weights()['origin_to_sub_kernels_of_layer%d' % l].grad.clone()
    torch.cuda.synchronize()
    rs, x64, w64, hk64, hb64 = _oracle(x, w, hk, hb, L, gs)
    close(s, rs, what='scores')
    for l in range(L):
        close(snap[l], w64['origin_to_sub_kernels_of_layer%d' % l].grad, what='dU_%d seen behind its event' % l)
    for k, p in cross.named_weights().items():
        close(p.grad, w64[k].grad, what=k)
    close(head.kernel.grad, hk64.grad, what='head kernel')


def test_fused_score_falls_back_for_other_shapes(dev):
    """Shapes outside the fused route (N*S not a multiple of 128, B not a multiple of 256, non-linear head) call the two layers."""
    from rec_now_amd.fused import dcn_mix_score, fused_route_available
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    x, xd, cross, head, w, hk, hb = _build(dev, 300, 96, 16, 3, 2, 9)
    assert not fused_route_available(cross, head, xd)
    gs = np.random.default_rng(3).normal(size=300).astype(np.float32)
    s = dcn_mix_score(cross, head, xd)
    s.backward(torch.from_numpy(gs).to(dev))
    rs, x64, w64, hk64, hb64 = _oracle(x, w, hk, hb, 2, gs)
    close(s, rs, what='scores')
    close(xd.grad, x64.grad, what='dx')
    x2, xd2, cross2, _, _, _, _ = _build(dev, 512, 256, 64, 2, 2, 10)
    head2 = MultiDenseLayer(1, 1, activation='tanh')
    head2(cross2(xd2[:256]))
    assert not fused_route_available(cross2, head2, xd2)
    assert dcn_mix_score(cross2, head2, xd2).shape == (512,)


def test_one_rank_rccl_group_equals_plain_path(dev):
    """The N > 1 path of bench.py on ONE GPU: a 1-rank RCCL group with every collective forced on (`dp.FORCE_COLLECTIVES`, what
    `bench.py --force-dist` sets) -- the layer-wise in-place all-reduce behind the per-layer events and the one-bucket form after
    the backward pass -- gives the loss and every gradient of the plain single-process path."""
    import os
    import socket
    import torch.distributed as dist
    from rec_now_amd import dp
    from rec_now_amd.fused import GpuEvent, dcn_mix_score, score_params
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss_fused
    B, D, S, N, L = 2048, 256, 64, 2, 3
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, S, N, L, 21)
    rng = np.random.default_rng(4)
    gd = torch.from_numpy(rng.integers(0, 40, B).astype(np.int64)).to(dev)
    yd = torch.from_numpy(rng.integers(0, 2, B).astype(np.float32)).to(dev)
    params = score_params(cross, head)

    def clear():
        for p in params:
            p.grad = None
        xd.grad = None

    def snap():
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params] + [xd.grad.detach().clone()]

    # the head-bias gradient is sum(d loss / d score) = 0 up to rounding: compare it on the scale of its terms (sum |ds| <= 1)
    scale_of = lambda r, p: 1.0 if p is head.bias else float(r.abs().max())     # noqa: E731

    # plain: normalise, then backward
    clear()
    ls, npair = pairwise_loss_fused(dcn_mix_score(cross, head, xd), yd, gd, reduce_mean=False)
    loss_bw, loss_plain, _ = dp.global_pairwise_loss(ls, npair)
    loss_bw.backward()
    plain = snap()
    assert float(npair) > 0 and abs(float(loss_plain) - np.log(2.0)) > 1e-3

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', device_id=dev)
    dp.FORCE_COLLECTIVES = True
    try:
        assert dp.is_dist()
        inv = 1.0 / (float(npair) + 1e-10)
        # (a) layer-wise: gradients written into per-layer buckets, all-reduced in place behind the layer's event
        events = [GpuEvent() for _ in range(L)]
        per_layer = lambda l: [cross.origin_to_sub_kernels[l], cross.sub_to_sub_kernels[l], cross.sub_to_origin_kernels[l],     # noqa: E731
                               cross.biases[l], cross.gate_layers[l].kernel]
        stages = [per_layer(L - 1) + [head.kernel, head.bias]] + [per_layer(l) for l in range(L - 2, -1, -1)]
        lw = dp.LayerwiseReducer(stages, [events[l] for l in range(L - 1, -1, -1)], dev)
        gbuf = [lw.buffer_of(p) for p in params]
        for rep in range(2):                           # twice: the buckets are reused step after step
            clear()
            ls, npair2 = pairwise_loss_fused(dcn_mix_score(cross, head, xd, layer_events=events, grad_buffers=gbuf), yd, gd,
                                             reduce_mean=False)
            lw.prepare(ls, npair2)
            ls.backward()
            loss_lw, p_glob = lw.reduce(ls, npair2)
            got = snap()
            assert lw.last_foreign == 0                # every gradient was produced inside its bucket
            assert float(p_glob) == float(npair)
            close(loss_lw, loss_plain, rtol=1e-6, what='loss (layer-wise reducer)')
            for g, r, p in zip(got[:-1], plain[:-1], params):
                close(g.reshape(-1), r.reshape(-1), rtol=2e-6, what='layer-wise reducer gradient', scale=scale_of(r, p))
            close(got[-1] * inv, plain[-1], rtol=2e-6, what='dx (unnormalised backward x 1/P)')
        # (b) one bucket after the backward pass
        clear()
        ls, npair3 = pairwise_loss_fused(dcn_mix_score(cross, head, xd), yd, gd, reduce_mean=False)
        ls.backward()
        loss_one, _ = dp.GradientAllReducer(params).all_reduce_with_loss(ls, npair3)
        got = snap()
        close(loss_one, loss_plain, rtol=1e-6, what='loss (one bucket)')
        for g, r, p in zip(got[:-1], plain[:-1], params):
            close(g.reshape(-1), r.reshape(-1), rtol=2e-6, what='one-bucket gradient', scale=scale_of(r, p))
    finally:
        dp.FORCE_COLLECTIVES = False
        dist.destroy_process_group()
