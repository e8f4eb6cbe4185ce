"""CPU: the oracle (oracle/dense_ref.py) against every golden the reference's own hot-path tests hold
(SURVEY.md section 4).  This is what pins the oracle; the GPU parity tests then compare HIP vs oracle."""
import numpy as np
import torch

import dense_ref as R

T = torch.from_numpy
TOL = 1e-5   # reference tolerance: sum|diff| < 1e-5 (util/numpy_tools.py metric)


def test_fm(golden):
    g = golden('fm')
    out = R.fm_layer([T(x) for x in g['inputs']])
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_dcn(golden):
    g = golden('dcn')
    out = R.dcn_layer(T(g['inputs']), [T(g['kernel_%d' % i]) for i in range(3)], [T(g['bias_%d' % i]) for i in range(3)])
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_multi_dense(golden):
    for name in ('multi_dense_2d', 'multi_dense_3d'):
        g = golden(name)
        out = R.multi_dense_layer(T(g['inputs']), T(g['kernel']), T(g['bias']))
        assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_multi_dense_wrong_leading_dim_raises():
    # tests/layers/test_multi_dense_layer.py:57-73
    try:
        R.multi_dense_layer(torch.zeros(4, 2, 4), torch.zeros(3, 4, 1))
    except ValueError:
        return
    raise AssertionError('expected an error')


def test_mmoe(golden):
    g = golden('mmoe')
    out = R.mmoe_layer(T(g['inputs']), [T(g['expert_kernel_0']), T(g['expert_kernel_1'])],
                       [T(g['expert_bias_0']), T(g['expert_bias_1'])], T(g['gate_kernel']), T(g['gate_bias']))
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_dcn_mix(golden):
    g = golden('dcn_mix')
    L = 2
    pick = lambda fmt: [T(g[fmt % l]) for l in range(L)]   # noqa: E731
    out = R.dcn_mix_layer(T(g['inputs']), pick('origin_to_sub_kernels_of_layer%d'), pick('sub_to_sub_kernels_of_layer%d'),
                          pick('sub_to_origin_kernels_of_layer%d'), pick('bias_of_layer%d'), pick('gate_of_layer%d'))
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_cin(golden):
    g = golden('cin')
    out = R.cin_layer([T(x) for x in g['inputs']], [T(g['weight_of_layer1']), T(g['weight_of_layer2'])], 10, 3)
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_cin_gemm_form_equals_line_by_line_form(golden):
    """oracle/dense_ref.cin_layer_gemm_form (what the full-size chunked oracle evaluates) against the line-by-line restatement:
    the reference's golden, every output mode, and fp64 gradients on a random case."""
    import torch
    g = golden('cin')
    ws = [T(g['weight_of_layer1']), T(g['weight_of_layer2'])]
    out = R.cin_layer_gemm_form([T(x) for x in g['inputs']], ws, 10, 3)
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL
    gen = torch.Generator().manual_seed(3)
    F, D, Hs, B = 5, 4, [6, 3, 7], 9
    xs = [torch.randn(B, D, generator=gen, dtype=torch.float64) for _ in range(F)]
    hp = [F] + Hs
    for oi in (True, False):
        for sc in (True, False):
            w_a = [torch.randn(1, 1, hp[k + 1], hp[k] * F, generator=gen, dtype=torch.float64).requires_grad_(True) for k in range(len(Hs))]
            w_b = [w.detach().clone().requires_grad_(True) for w in w_a]
            xa = [x.clone().requires_grad_(True) for x in xs]
            xb = [x.clone().requires_grad_(True) for x in xs]
            ya, yb = R.cin_layer(xa, w_a, F, D, oi, sc), R.cin_layer_gemm_form(xb, w_b, F, D, oi, sc)
            assert ya.shape == yb.shape and float((ya - yb).abs().max()) < 1e-12
            gy = torch.randn(ya.shape, generator=gen, dtype=torch.float64)
            ya.backward(gy)
            yb.backward(gy)
            for a, b in zip(xa + w_a, xb + w_b):
                assert float((a.grad - b.grad).abs().max()) < 1e-11


def ple_layers_from_fixture(g, to=T):
    layers = []
    for li in range(3):
        layer = {'dnn': [], 'gate': []}
        for gi in range(3):
            layer['dnn'].append([(to(g['l%d_g%d_dnn%d_kernel' % (li, gi, di)]), to(g['l%d_g%d_dnn%d_bias' % (li, gi, di)]))
                                 for di in range(2)])
            key = 'l%d_g%d_gate_kernel' % (li, gi)
            layer['gate'].append((to(g[key]), to(g['l%d_g%d_gate_bias' % (li, gi)])) if key in g else None)
        layers.append(layer)
    return layers


def test_ple(golden):
    g = golden('ple')
    out = R.ple_layer(T(g['inputs']), ple_layers_from_fixture(g), [True, False, False])
    assert R.calc_sum_of_abs_diff(out[0].numpy(), g['golden_task1']) < TOL
    assert R.calc_sum_of_abs_diff(out[1].numpy(), g['golden_task2']) < TOL


# ---- literal-input goldens (no RNG) -------------------------------------------------------------------
def _pairwise_case():
    g = torch.tensor([1, 1, 2, 2, 2.]).reshape(-1, 1)
    s = torch.tensor([0, 1, 2, 3, 4.]).reshape(-1, 1)
    y = torch.tensor([1.1, 0, 0, 1, 1]).reshape(-1, 1)
    return g, s, y


def test_occurance_power_weight():
    # tests/rec_block/test_pairwise_loss_from_batch.py:19-31
    ids = [1, 1, 2, 4, 4, 4]
    np.testing.assert_allclose(R.occurance_power_weight(ids, -1).numpy(), [.5, .5, 1, 1 / 3, 1 / 3, 1 / 3], atol=1e-4)
    np.testing.assert_allclose(R.occurance_power_weight(ids, 2).numpy(), [4, 4, 1, 9, 9, 9], atol=1e-4)


def test_pairwise_goldens():
    # tests/rec_block/test_pairwise_loss_from_batch.py:33-74
    g, s, y = _pairwise_case()
    f = lambda p, n, w: R.bpr_loss_func(p, n, w, 1.0)   # noqa: E731
    assert abs(R.pairwise_loss(s, y, g, f, click_occurance_power=-0.5).item() - 0.5415076) < 1e-4
    wf = lambda a, b, **k: (a > b).float()              # noqa: E731
    assert abs(R.pairwise_loss(s, y, g, f, click_occurance_power=-0.5, label_pair_to_weight_func=wf).item() - 0.5415076) < 1e-4
    m = torch.tensor([True, True, False, False, False]).reshape(-1, 1)
    assert abs(R.pairwise_loss(s, y, g, f, click_occurance_power=-0.5, mask=m).item() - 1.3132617) < 1e-4


def test_listwise_goldens():
    # tests/rec_block/test_listwise_loss_from_batch.py:18-51
    g = torch.tensor([1, 1, 2, 1, 2, 2, 3, 4.])
    y = torch.tensor([1, 1, 1, 0, 0, 0, 1, 0.])
    s = torch.tensor([.1, .01, .2, .001, .02, .002, .3, .4])
    m, lab, lg = R.to_listwise_sample(g, y, s)
    assert lab.shape[0] == 2
    assert abs(R.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg).item() - 1.0291535) < 1e-4
    m, lab, lg = R.to_listwise_sample(torch.tensor([3., 4.]), torch.tensor([1., 0.]), torch.tensor([.3, .4]))
    assert lab.shape[0] == 0
    assert abs(R.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg).item()) < 1e-4


def test_pair_order_is_row_major():
    g = torch.tensor([7., 3., 7., 3., 7.])
    y = torch.tensor([1., 0., 0., 1., 2.])
    pos, neg = R.pair_indices(y, g)
    assert pos.tolist() == [0, 3, 4, 4] and neg.tolist() == [2, 1, 0, 2]


# ---- the plain-C restatement (oracle/pairs_oracle.c) is pinned to the same goldens and to the dense oracle ----------------
def test_c_oracle_reference_goldens():
    import pairs_oracle as C
    g = [1, 1, 2, 2, 2]
    s = [0, 1, 2, 3, 4]
    y = [1.1, 0, 0, 1, 1]
    loss, _, P = C.pairwise_bpr(g, y, s, power=-0.5)
    assert P == 3 and abs(loss - 0.5415076) < 1e-4
    loss, _, _ = C.pairwise_bpr(g, y, s, mask=[1, 1, 0, 0, 0], power=-0.5)
    assert abs(loss - 1.3132617) < 1e-4


def test_c_oracle_equals_dense_oracle():
    import pairs_oracle as C
    rng = np.random.default_rng(0)
    B = 700
    g = rng.integers(0, 11, B).astype(np.float32)
    y = rng.integers(0, 3, B).astype(np.float32)
    s = rng.normal(size=B).astype(np.float32)
    m = rng.random(B) < 0.8
    for wrong in (False, True):
        pos, neg = C.pair_indices(g, y, s, m, flags=1 | (2 if wrong else 0))
        rpos, rneg = R.pair_indices(torch.from_numpy(y), torch.from_numpy(g), wrong, torch.from_numpy(s), torch.from_numpy(m))
        assert np.array_equal(pos, rpos.numpy()) and np.array_equal(neg, rneg.numpy())
    s64 = torch.from_numpy(s).double().requires_grad_(True)
    rl = R.pairwise_loss(s64, torch.from_numpy(y).double(), torch.from_numpy(g), click_occurance_power=-0.5, mask=torch.from_numpy(m))
    rl.backward()
    loss, d, _ = C.pairwise_bpr(g, y, s, m, power=-0.5)
    assert abs(loss - rl.item()) < 1e-9 * max(1, abs(rl.item())) + 1e-7
    assert np.abs(d - s64.grad.numpy()).max() < 1e-7


# ---- SURVEY section 8f rows -----------------------------------------------------------------------------------------
def test_inner_pnn(golden):
    # /root/reference/tests/layers/test_inner_pnn_layer.py:18-39
    g = golden('inner_pnn')
    out = R.inner_pnn_layer([T(x) for x in g['inputs']])
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_senet(golden):
    # /root/reference/tests/layers/test_senet_layer.py:18-39
    g = golden('senet')
    out = R.senet_layer([T(g['input_%d' % i]) for i in range(3)], [T(g['dense_0_kernel']), T(g['dense_1_kernel'])],
                        [T(g['dense_0_bias']), T(g['dense_1_bias'])])
    assert R.calc_sum_of_abs_diff(out.numpy(), g['golden']) < TOL


def test_focal_loss_literals():
    # /root/reference/tests/rec_block/test_focal_loss.py:17-28
    labels = torch.tensor([1, 1, 0, 0], dtype=torch.float32).reshape(-1, 1)
    logits = torch.tensor([0.9, 0.8, 0.7, 0.6], dtype=torch.float32).reshape(-1, 1)
    assert abs(float(R.focal_crossentropy_loss(labels, logits, alpha=None, gamma=None)) - 0.71323216) < 1e-5
    assert abs(float(R.focal_crossentropy_loss(labels, logits, alpha=0.25, gamma=None)) - 0.44589227) < 1e-5
    assert abs(float(R.focal_crossentropy_loss(labels, logits, alpha=None, gamma=1)) - 0.40516436) < 1e-5


ATTN_USER = [[[0.1, 0.2], [-0.1, -0.2]], [[0.3, 0.4], [-0.3, -0.4]]]
ATTN_DOC = [[0.1, 0.2], [0.3, 0.4]]
ATTN_GOLDEN = {False: ([[0.01, 0.02], [0.15, 0.2]], [[0.0], [0.0]]), True: ([[0.005, 0.01], [0.075, 0.1]], [[0.05], [0.25]])}


def test_attention_by_dot_product_literals():
    # /root/reference/tests/rec_block/test_attention.py:19-56
    for filter_neg, (gm, gs) in ATTN_GOLDEN.items():
        mat, score = R.attention_by_dot_product(torch.tensor(ATTN_USER), torch.tensor(ATTN_DOC), filter_neg=filter_neg)
        assert R.calc_sum_of_abs_diff(mat.numpy(), gm) < TOL
        assert R.calc_sum_of_abs_diff(score.numpy(), gs) < TOL


def embed_pool_case():
    """/root/reference/tests/rec_block/test_embedding_util.py:71-109 (inputs and the two expected results)."""
    import numpy as np
    params = np.array([[i, -i] for i in range(40)], np.float32)
    ids = np.array([[0, 10, 20, 30], [21, 30, 31, 1]], np.int64)
    slots = ((ids.astype(np.float64) + 0.5) / 10.0).astype(np.int32)
    weights = ids.astype(np.float32) * 10.0
    exp_w = [[[1000., -1000.], [9000., -9000.]], [[0., 0.], [18610., -18610.]]]
    exp_n = [[[10., -10.], [30., -30.]], [[0., 0.], [61., -61.]]]
    return params, ids, slots, [1, 3], weights, exp_w, exp_n


def test_embedding_pool_literals():
    import numpy as np
    params, ids, slots, targets, weights, exp_w, exp_n = embed_pool_case()
    out = R.embedding_using_sparse_batch_segment_ids(T(params), T(slots), targets, T(ids), weights=T(weights))
    assert np.array_equal(out.numpy(), np.array(exp_w, np.float32))
    out = R.embedding_using_sparse_batch_segment_ids(T(params), T(slots), targets, T(ids))
    assert np.array_equal(out.numpy(), np.array(exp_n, np.float32))


def test_grouped_c_port_equals_the_quadratic_restatement():
    """oracle_pairwise_bpr_grouped (the segment-based CPU port bench.py times at B = 65536) against the O(B^2) restatement of
    the reference formulation: same pair count, loss and gradient, incl. NaN / inf / signed-zero group ids, masks, 3 label levels,
    wrong-order pairs and occurrence weights."""
    import pairs_oracle as PO
    rng = np.random.default_rng(0)
    for B, G in ((3000, 40), (5000, 7), (2000, 2000), (1, 1), (0, 1)):
        g = rng.integers(0, G, B).astype(np.float32)
        if B > 100:
            g[::97] = np.nan
            g[5], g[6], g[7] = -0.0, 0.0, np.inf
        y = rng.integers(0, 3, B).astype(np.float32)
        s = rng.normal(size=B).astype(np.float32)
        m = (rng.random(B) < 0.9).astype(np.uint8)
        for flags, power in ((1, 0.0), (3, -0.5), (1, 1.0)):
            a = PO.pairwise_bpr(g, y, s, m, flags, 1.3, power)
            b = PO.pairwise_bpr(g, y, s, m, flags, 1.3, power, grouped=True)
            assert a[2] == b[2]
            assert abs(a[0] - b[0]) <= 1e-12 * max(1.0, abs(a[0]))
            assert B == 0 or np.abs(a[1] - b[1]).max() <= 1e-12


def test_c_oracle_under_sanitizers():
    """SURVEY.md section 5 (sanitizer build): oracle/pairs_oracle.c + oracle/asan_driver.c compiled with -fsanitize=address,undefined and run on
    the reference's literal goldens, the empty / one-row / NaN-id edge cases and seeded random batches (`make -C oracle asan`)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(['make', '-C', os.path.join(root, 'oracle'), '-s', 'asan'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'asan driver ok' in out.stdout


def test_gemm_dispatch_logic_under_sanitizers():
    """SURVEY.md section 5 (sanitizer build), host side: the GEMM's tile-family / small-M / split-K / slab-workspace logic
    (rec_now_amd/csrc/gemm_dispatch.hpp, host-only) swept over shapes by tools/san/dispatch_san.cpp under -fsanitize=address,undefined."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(['make', '-C', os.path.join(root, 'tools', 'san'), '-s', 'san'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'dispatch sanitizer driver ok' in out.stdout
