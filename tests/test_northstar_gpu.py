"""GPU parity of the north-star path at BASELINE.json's own sizes, EVERY output and gradient against the fp64 oracle.

configs[2] = DCNMixLayer(3 cross layers, low-rank 64, 2 experts) + MultiDenseLayer(1,1) head + in-batch pairwise loss at
B = 65536, D = 64 x 16: at this size the library dispatches to the exact-128 formulation, the persistent short-K products and the
split-K weight-gradient products (B >= 32768), none of which the small-shape sweeps of test_layers_gpu.py reach.  The oracle
(oracle/dense_ref.py, reference /root/reference/rec_now/layers/dcn_mix_layer.py:135-150, multi_dense_layer.py:80-94,
rec_block/pairwise_loss_from_batch.py:228-279) is row-separable, so it is evaluated chunk-wise in fp64 with the weight
gradients summed over the chunks (tests/_chunked_oracle.py); the O(B^2) pair part runs in the plain-C restatement
(oracle/pairs_oracle.c), which needs no (B,B) temporaries.  Tolerance: 1e-5 of the largest reference magnitude (north_star).
configs[1] = pairwise_loss_from_batch at B = 8192 with ~128 user groups: pair list bit-exact, loss and gradient vs the C oracle."""
import numpy as np
import pytest
import torch

import dense_ref as R
import pairs_oracle as PO
from _chunked_oracle import GEMM_PRECISIONS, close, gemm_precision, run_chunked, weights64

pytestmark = pytest.mark.gpu


def _mix_fwd(w64, L, head=None):
    pick = lambda fmt: [w64[fmt % l] for l in range(L)]     # noqa: E731

    def fwd(xc):
        y = R.dcn_mix_layer(xc, pick('origin_to_sub_kernels_of_layer%d'), pick('sub_to_sub_kernels_of_layer%d'),
                            pick('sub_to_origin_kernels_of_layer%d'), pick('bias_of_layer%d'), pick('gate_of_layer%d/kernel'),
                            'tanh', 'tanh')
        if head is not None:
            return R.multi_dense_layer(y, w64['head/kernel'], w64['head/bias']).reshape(-1)
        return y
    return fwd


def _randomise(layer, seed, bias_scale=0.1):
    """Glorot weights as built, but non-zero biases (zero biases would hide errors in the bias rows of the packed products)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    with torch.no_grad():
        for name, p in layer.named_weights().items():
            if 'bias' in name:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).mul_(bias_scale).to(p.device))


def test_dcn_mix_config3_every_gradient_vs_oracle(dev):
    """configs[2] layer, B = 65536, D = 1024, S = 64, N = 2, L = 3: y, dx (all rows) and all 15 weight gradients vs fp64."""
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    B, D, S, N, L = 65536, 1024, 64, 2, 3
    g = torch.Generator(device='cpu').manual_seed(21)
    x = torch.randn(B, D, generator=g) * 0.5
    gy = torch.randn(B, D, generator=g)
    layer = DCNMixLayer(S, num_layer=L, num_expert=N)
    xd = x.to(dev).requires_grad_(True)
    layer(xd[:256])
    _randomise(layer, 22)
    w64 = weights64(layer.named_weights())
    (ry,), rdx, rgrads = run_chunked(_mix_fwd(w64, L), x, gy, w64, chunk=4096)
    assert len(rgrads) == 5 * L
    for prec in GEMM_PRECISIONS:            # exact fp32 MFMA and the six-term 3 x bf16 split: ONE oracle, the same bound
        xd.grad = None
        for p in layer.named_weights().values():
            p.grad = None
        with gemm_precision(prec):
            y = layer(xd)
            y.backward(gy.to(dev))
            torch.cuda.synchronize()
        close(y, ry, what='y ' + prec)
        close(xd.grad, rdx, what='dx ' + prec)
        for name, p in layer.named_weights().items():
            close(p.grad, rgrads[name], what=name + ' ' + prec)


# (the shard sizes: tests/test_step_gpu.py, through the reducer.  The fused node at the FULL batch is what tests/test_step_gpu.py
# ::test_step_route_at_the_metric_batch_vs_oracle runs -- the step entry launches the same kernels, held bit for bit to the autograd route of the fused node
# at smaller batches -- so the fused route is compared here at half the batch: the suite has to fit the driver's time limit)
@pytest.mark.parametrize('B,route', [(65536, 'layers'), (32768, 'fused')])
def test_config3_model_drop_in_signature(dev, B, route):
    """configs[2] end to end through the drop-in signatures: DCNMixLayer -> MultiDenseLayer(1,1) -> pairwise_loss(outputs, labels,
    groups): loss, pair count, d loss / d x and every weight gradient (cross layers and head) vs the oracle.
    route 'fused': the same two layers evaluated by the model-level fused node (rec_now_amd/fused.py)."""
    from rec_now_amd.fused import dcn_mix_score, fused_route_available
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss
    D, S, N, L = 1024, 64, 2, 3
    rng = np.random.default_rng(31 + B)
    x = rng.normal(0.0, 0.7, (B, D)).astype(np.float32)
    groups = rng.integers(0, B // 64, B).astype(np.float32)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    cross, head = DCNMixLayer(S, num_layer=L, num_expert=N), MultiDenseLayer(1, 1)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    head(cross(xd[:256]))
    _randomise(cross, 32)
    with torch.no_grad():
        head.kernel.mul_(40.0)          # scores of O(1): pair terms away from the softplus(0) = ln 2 plateau
        head.bias.fill_(0.3)
    named = dict(cross.named_weights())
    named['head/kernel'], named['head/bias'] = head.kernel, head.bias
    w64 = weights64(named)
    fwd = _mix_fwd(w64, L, head=True)
    (rs,), _, _ = run_chunked(fwd, torch.from_numpy(x), None, w64, chunk=4096, want_dx=False)
    assert np.abs(rs).max() > 0.5 and np.abs(rs).std() > 0.05            # not the degenerate all-scores-equal case
    rloss, rds, rP = PO.pairwise_bpr(groups, labels, rs.astype(np.float32))
    assert rP > B and abs(rloss - np.log(2.0)) > 1e-3
    _, rdx, rgrads = run_chunked(fwd, torch.from_numpy(x), torch.from_numpy(rds), w64, chunk=4096)
    for prec in GEMM_PRECISIONS:            # both arithmetics of the products against the ONE oracle evaluation above
        xd.grad = None
        for p in named.values():
            p.grad = None
        with gemm_precision(prec):
            if route == 'fused':
                assert fused_route_available(cross, head, xd)
                scores = dcn_mix_score(cross, head, xd)
            else:
                scores = head(cross(xd)).reshape(-1)
            loss, n_pair = pairwise_loss(scores, torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev), return_num_pair=True)
            loss.backward()
            torch.cuda.synchronize()
        close(scores, rs, what='scores ' + prec)
        assert int(n_pair.item()) == rP
        close(loss, np.float64(rloss), what='loss ' + prec)
        close(xd.grad, rdx, what='dx ' + prec)
        for name, p in named.items():
            # d loss / d head bias = sum_i dscore_i, which is 0 in exact arithmetic (every pair adds +t to one row and -t to another):
            # its error is measured against the magnitude of the terms that cancel
            close(p.grad, rgrads[name], what=name + ' ' + prec, scale=np.abs(rds).sum() if name == 'head/bias' else None)


def test_config2_pairwise_literal_case(dev):
    """configs[1]: B = 8192, integers(0, 128) user groups, scalar scores: pair list bit-exact in the reference's order
    (np.nonzero of the row-major (B,B) mask, rec_block/pairwise_loss_from_batch.py:217,272-273), loss and gradient."""
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pair_indices, pairwise_loss
    B = 8192
    rng = np.random.default_rng(2)
    groups = rng.integers(0, 128, B).astype(np.float32)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    scores = rng.normal(size=B).astype(np.float32)
    sd = torch.from_numpy(scores).to(dev).requires_grad_(True)
    yd, gd = torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev)
    pos, neg = pair_indices(sd, yd, gd)
    rpos, rneg = PO.pair_indices(groups, labels, scores)
    assert np.array_equal(pos.cpu().numpy(), rpos) and np.array_equal(neg.cpu().numpy(), rneg)
    # the dense torch restatement agrees with the C restatement here (B^2 = 67 M mask elements still fit)
    dpos, dneg = R.pair_indices(torch.from_numpy(labels), torch.from_numpy(groups))
    assert np.array_equal(np.asarray(dpos), rpos) and np.array_equal(np.asarray(dneg), rneg)
    loss, n_pair = pairwise_loss(sd.reshape(-1, 1), yd.reshape(-1, 1), gd.reshape(-1, 1), return_num_pair=True)
    loss.backward()
    rloss, rds, rP = PO.pairwise_bpr(groups, labels, scores)
    assert int(n_pair.item()) == rP == len(rpos)
    close(loss, np.float64(rloss), what='loss')
    close(sd.grad, rds, what='dscores')
    # wrong-order pairs only + occurrence weights, same batch
    sd.grad = None
    loss2 = pairwise_loss(sd, yd, gd, only_use_wrong_order_pair=True, click_occurance_power=-0.5)
    loss2.backward()
    rloss2, rds2, _ = PO.pairwise_bpr(groups, labels, scores, flags=3, power=-0.5)
    close(loss2, np.float64(rloss2), what='loss (wrong order, occurrence weights)')
    close(sd.grad, rds2, what='dscores (wrong order, occurrence weights)')


def _ple_oracle_fwd(layer, w64, dims, n_groups, is_shared):
    """The fp64 oracle of a PLELayer over the fp64 leaves `w64`, weights mapped by their reference names."""
    def fwd(xc):
        layers = []
        for l in range(len(dims)):
            entry = {'dnn': [], 'gate': []}
            for gi in range(n_groups):
                tn = layer.task_names[gi]
                scope = 'PLE/ple_layer_%d/task_%s' % (l, tn)
                entry['dnn'].append([(w64['%s/%s/MultiDenseLayer_%d/kernel' % (scope, scope, i)],
                                      w64['%s/%s/MultiDenseLayer_%d/bias' % (scope, scope, i)]) for i in range(len(dims[l]))])
                gk = 'PLE/ple_gate_%d/task_%s/dense/kernel' % (l, tn)
                entry['gate'].append((w64[gk], w64[gk[:-6] + 'bias']) if gk in w64 else None)
            layers.append(entry)
        return R.ple_layer(xc, layers, is_shared, activation='tanh')
    return fwd


def test_ple_config5_per_rank_every_gradient_vs_oracle(dev):
    """configs[4] layer alone: PLELayer, 3 tasks + 1 shared group, D_in = 128 x 32 = 4096: task outputs, dx and
    every expert / gate weight gradient vs the fp64 oracle (/root/reference/rec_now/layers/ple_layer.py:295-321), weights mapped
    by their reference names."""
    from rec_now_amd.layers.ple_layer import PLELayer
    # (8192 rows here: the SAME layer at the per-rank size B = 32 768, every expert / gate gradient by reference names, is part of
    # test_config5_end_to_end_per_rank_size_vs_oracle below -- one fp64 oracle of 32 768 x 4096 per suite run is what the time limit allows)
    B, Din = 8192, 4096
    dims, n_exp = [[512, 256], [256, 128]], 2
    g = torch.Generator(device='cpu').manual_seed(41)
    x = torch.randn(B, Din, generator=g) * 0.05
    layer = PLELayer(3, dims, n_exp, 1, activation='tanh', name='PLE')
    xd = x.to(dev).requires_grad_(True)
    layer(xd[:256])
    _randomise(layer, 42)
    outs = layer(xd)
    gys = [torch.randn(B, o.shape[1], generator=g) for o in outs]
    torch.autograd.backward(list(outs), [v.to(dev) for v in gys])
    named = dict(layer.named_weights())
    w64 = weights64(named)
    n_groups = len(layer.task_names[:layer.num_total_task])
    is_shared = layer.is_shared_tasks

    fwd = _ple_oracle_fwd(layer, w64, dims, n_groups, is_shared)

    routs, rdx, rgrads = run_chunked(fwd, x, gys, w64, chunk=2048)
    assert len(routs) == len(outs) == 3
    for t, (o, ro) in enumerate(zip(outs, routs)):
        close(o, ro, what='task %d output' % t)
    close(xd.grad, rdx, what='dx')
    checked = 0
    for name, p in named.items():
        if p.grad is None:
            assert np.abs(rgrads[name]).max() == 0.0, name
            continue
        close(p.grad, rgrads[name], what=name)
        checked += 1
    assert checked >= 4 * 2 * 2 * 2 + 4 + 3          # expert kernels/biases of both layers + their gates


def test_config5_end_to_end_per_rank_size_vs_oracle(dev):
    """configs[4] END TO END at the per-rank size of its 8-GPU row: x (32 768, 4096) -> PLELayer(3 tasks + 1 shared group,
    [[512, 256], [256, 128]], 2 experts) -> 3 heads MultiDenseLayer(1, 3) -> listwise loss on task 0 (512 lists) (+ small pointwise terms on
    the task logits so that every expert carries a gradient): loss, number of valid lists, d loss / d x, the head and EVERY expert / gate
    gradient against the fp64 oracle -- layers chunk-wise, the listwise stage in the reference's dense (G, B) form
    (/root/reference/rec_now/layers/ple_layer.py:295-321, rec_block/listwise_loss_from_batch.py:89-173)."""
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    from rec_now_amd.layers.ple_layer import PLELayer
    from rec_now_amd.rec_block.listwise_loss_from_batch import listwise_loss_from_batch
    B, Din, G = 32768, 4096, 512
    dims, n_exp = [[512, 256], [256, 128]], 2
    gen = torch.Generator(device='cpu').manual_seed(55)
    rng = np.random.default_rng(55)
    x = torch.randn(B, Din, generator=gen) * 0.3
    groups = rng.integers(0, G, B).astype(np.float32)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    layer, head = PLELayer(3, dims, n_exp, 1, activation='tanh', name='PLE'), MultiDenseLayer(1, 3)
    xd = x.to(dev).requires_grad_(True)
    head(torch.stack(list(layer(xd[:256]))))
    _randomise(layer, 56)
    with torch.no_grad():
        head.kernel.mul_(6.0)                    # logits of O(1): the softmax over a list is far from uniform
        head.bias.fill_(0.1)
    gd, yd = torch.from_numpy(groups).to(dev), torch.from_numpy(labels).to(dev)
    logits = head(torch.stack(list(layer(xd)))).reshape(3, B)
    loss, nv = listwise_loss_from_batch(gd, yd, logits[0], return_num_list=True)
    # (+ pointwise terms on all three logits: the listwise gradient sums to zero inside every list, so bias gradients made of it alone are sums that
    # cancel to ~1e-3 of their terms and a relative bound on them measures the conditioning of the sum, not the kernels)
    total = loss + (0.5 * logits[0].sum() + 0.01 * (logits[1].sum() - logits[2].sum())) / B
    total.backward()
    named = dict(layer.named_weights())
    named['head/kernel'], named['head/bias'] = head.kernel, head.bias
    w64 = weights64(named)
    ple = _ple_oracle_fwd(layer, w64, dims, len(layer.task_names[:layer.num_total_task]), layer.is_shared_tasks)
    fwd = lambda xc: R.multi_dense_layer(torch.stack(list(ple(xc))), w64['head/kernel'], w64['head/bias']).reshape(3, -1).t()      # noqa: E731  (rows, 3)
    (rl,), _, _ = run_chunked(fwd, x, None, w64, chunk=2048, want_dx=False)
    assert rl[:, 0].std() > 0.1                 # the softmax over a list is far from uniform
    l0 = torch.from_numpy(rl[:, 0].copy()).requires_grad_(True)
    _, lab, lg = R.to_listwise_sample(torch.from_numpy(groups), torch.from_numpy(labels).double(), l0)
    rloss = R.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg)
    rloss.backward()
    assert int(nv.item()) == lab.shape[0] > G // 2
    close(loss, rloss.detach(), what='listwise loss')
    close(logits.t(), rl, what='task logits')
    gl = torch.stack([l0.grad + 0.5 / B, torch.full((B,), 0.01 / B, dtype=torch.float64), torch.full((B,), -0.01 / B, dtype=torch.float64)], dim=1)
    _, rdx, rgrads = run_chunked(fwd, x, gl, w64, chunk=2048)
    close(xd.grad, rdx, what='dx')
    checked = 0
    for name, p in named.items():
        if p.grad is None:
            assert np.abs(rgrads[name]).max() == 0.0, name
            continue
        close(p.grad, rgrads[name], what=name)
        checked += 1
    assert checked >= 4 * 2 * 2 * 2 + 4 + 3 + 2       # experts of both layers + gates + the heads
