"""GPU parity for the GEMM-backed layers: MultiDense, MMoE, PLE, DCN, DCNMix and the raw GEMM entry point.
Each layer is checked (a) against the reference's own golden (fixtures regenerated from the reference tests' seeds),
(b) forward + backward against the oracle (fp64 autograd of the dense restatement) on seeded random inputs.
Tolerance: 1e-5 relative to the largest magnitude of the compared tensor (north_star: 1e-5 rel fp32)."""
import ctypes

import os

import numpy as np
import pytest
import torch

import dense_ref as R
from test_oracle_golden import ple_layers_from_fixture

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def close(a, b, rtol=RTOL):
    a = a.detach().cpu().double().numpy() if hasattr(a, 'detach') else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if hasattr(b, 'detach') else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    if os.environ.get('RECNOW_TEST_MARGIN_LOG') and err > 0.3 * rtol * scale:      # diagnostics: comparisons that use more than 30 % of their bound
        with open(os.environ['RECNOW_TEST_MARGIN_LOG'], 'a') as fh:
            fh.write('%.3f of the bound  %s\n' % (err / (rtol * scale), os.environ.get('PYTEST_CURRENT_TEST', '')))
    assert err <= rtol * scale, 'max err %.3g vs scale %.3g (rel %.3g)' % (err, scale, err / scale)


# ---- raw GEMM ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,ta,tb', [(1, 1, 1, 0, 0), (5, 3, 7, 0, 0), (130, 33, 70, 0, 1), (257, 129, 65, 1, 0),
                                         (300, 144, 1024, 0, 0), (64, 64, 5000, 1, 0), (1000, 160, 130, 0, 1),
                                         (130, 1024, 3000, 1, 0)])
def test_gemm_vs_fp64(dev, M, N, K, ta, tb):
    from rec_now_amd import _lib
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32)
    Bm = rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    Ad, Bd, bd = (torch.from_numpy(v).to(dev) for v in (A, Bm, bias))
    C = torch.empty((M, N), device=dev)
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_trans = Ad.data_ptr(), A.shape[1], ta
    d.B, d.ldb, d.b_trans = Bd.data_ptr(), Bm.shape[1], tb
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.bias, d.act = bd.data_ptr(), 2     # tanh epilogue
    lib = _lib.load()
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    A64 = A.astype(np.float64).T if ta else A.astype(np.float64)
    B64 = Bm.astype(np.float64).T if tb else Bm.astype(np.float64)
    pre = A64 @ B64 + bias
    ref = np.tanh(pre)
    # error budget: fp32 accumulation error of the pre-activation (~1e-7 * sum|a*b|) through tanh' <= 1
    bound = 2e-7 * (np.abs(A64) @ np.abs(B64)).max() + 1e-6
    assert np.abs(C.cpu().numpy() - ref).max() <= bound


# ---- MultiDense ------------------------------------------------------------------------------------------------
def test_multi_dense_reference_goldens(dev, golden):
    # /root/reference/tests/layers/test_multi_dense_layer.py:19-55
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    for name in ('multi_dense_2d', 'multi_dense_3d'):
        g = golden(name)
        x = torch.from_numpy(g['inputs']).to(dev)
        dense_layer = MultiDenseLayer(1, 3)
        dense_layer(x)                                  # builds
        dense_layer.set_weights_by_name({'kernel': g['kernel'], 'bias': g['bias']})
        result = dense_layer(x)
        assert calc_sum_of_abs_diff(result, g['golden']) < 1e-5


def test_multi_dense_wrong_3d_raises(dev):
    # /root/reference/tests/layers/test_multi_dense_layer.py:57-73 (InvalidArgumentError there)
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    with pytest.raises(ValueError, match=r'\[4, 2, 4\] vs. \[3, 4, 1\]'):
        MultiDenseLayer(1, 3)(torch.zeros(4, 2, 4, device=dev))


@pytest.mark.parametrize('B,D,U,N,batched,act', [(7, 5, 3, 2, False, None), (300, 64, 48, 3, True, 'relu'),
                                                  (1000, 130, 1, 1, False, 'tanh'), (513, 256, 200, 4, False, 'sigmoid'),
                                                  (2048, 1024, 1, 1, False, None)])
def test_multi_dense_fwd_bwd_vs_oracle(dev, B, D, U, N, batched, act):
    from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
    rng = np.random.default_rng(B + D + U)
    x = rng.normal(0, 0.5, (N, B, D) if batched else (B, D)).astype(np.float32)
    k = rng.uniform(-0.3, 0.3, (N, D, U)).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, (N, 1, U)).astype(np.float32)
    gy = rng.normal(size=(N, B, U)).astype(np.float32)
    layer = MultiDenseLayer(U, N, activation=act)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    layer(xd)
    layer.set_weights_by_name({'kernel': k, 'bias': b})
    y = layer(xd)
    y.backward(torch.from_numpy(gy).to(dev))
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    k64 = torch.from_numpy(k).double().requires_grad_(True)
    b64 = torch.from_numpy(b).double().requires_grad_(True)
    ry = R.multi_dense_layer(x64, k64, b64, act)
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry)
    close(xd.grad, x64.grad)
    close(layer.kernel.grad, k64.grad)
    close(layer.bias.grad, b64.grad)


# ---- MMoE --------------------------------------------------------------------------------------------------------
def _mmoe_weights(g):
    return {'MMoE/experts/MultiDenseLayer_0/kernel': g['expert_kernel_0'], 'MMoE/experts/MultiDenseLayer_0/bias': g['expert_bias_0'],
            'MMoE/experts/MultiDenseLayer_1/kernel': g['expert_kernel_1'], 'MMoE/experts/MultiDenseLayer_1/bias': g['expert_bias_1'],
            'MMoE/gates/MultiDenseLayer/kernel': g['gate_kernel'], 'MMoE/gates/MultiDenseLayer/bias': g['gate_bias']}


def test_mmoe_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_mmoe_layer.py:18-38
    from rec_now_amd.layers.mmoe_layer import MMOELayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('mmoe')
    inputs = torch.from_numpy(g['inputs']).to(dev)
    mmoe_layer = MMOELayer(2, 4, [8, 3], name="MMoE")
    mmoe_layer(inputs, False)
    mmoe_layer.set_weights_by_name(_mmoe_weights(g))
    result = mmoe_layer(inputs, False)
    assert isinstance(result, list) and len(result) == 2
    assert calc_sum_of_abs_diff(torch.stack(result), g['golden']) < 1e-5
    merged = mmoe_layer(inputs)                 # merge_output=True
    assert merged.shape == (2, 3, 3)


@pytest.mark.parametrize('act', [None, 'relu'])
def test_mmoe_fwd_bwd_vs_oracle(dev, act):
    from rec_now_amd.layers.mmoe_layer import MMOELayer
    rng = np.random.default_rng(11)
    B, D, T, N, dims = 700, 96, 3, 5, [64, 32]
    x = rng.normal(0, 1, (B, D)).astype(np.float32)
    layer = MMOELayer(T, N, dims, activation=act, name='m')
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    layer(xd)
    w = {k: rng.uniform(-0.2, 0.2, tuple(v.shape)).astype(np.float32) for k, v in layer.named_weights().items()}
    layer.set_weights_by_name(w)
    gy = rng.normal(size=(T, B, dims[-1])).astype(np.float32)
    y = layer(xd)
    y.backward(torch.from_numpy(gy).to(dev))
    w64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    ry = R.mmoe_layer(x64, [w64['m/experts/MultiDenseLayer_%d/kernel' % i] for i in range(2)],
                      [w64['m/experts/MultiDenseLayer_%d/bias' % i] for i in range(2)],
                      w64['m/gates/MultiDenseLayer/kernel'], w64['m/gates/MultiDenseLayer/bias'], activation=act)
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry)
    close(xd.grad, x64.grad)
    for k, p in layer.named_weights().items():
        close(p.grad, w64[k].grad)


# ---- PLE ---------------------------------------------------------------------------------------------------------
def _ple_names(li, gi, names=('shared_0', 'special_0', 'special_1')):
    # the reference names groups shared_0.., special_0.. but assigns them through zip() -> group gi gets task_names[gi]
    task_names = ['shared_0', 'shared_1', 'special_0', 'special_1']     # reference :108-111 with num_task = 2
    return task_names[gi]


def _load_ple(layer, g):
    vals = {}
    for li in range(3):
        for gi in range(3):
            tn = _ple_names(li, gi)
            for di in range(2):
                base = 'PLE/ple_layer_%d/task_%s/PLE/ple_layer_%d/task_%s/MultiDenseLayer_%d/' % (li, tn, li, tn, di)
                vals[base + 'kernel'] = g['l%d_g%d_dnn%d_kernel' % (li, gi, di)]
                vals[base + 'bias'] = g['l%d_g%d_dnn%d_bias' % (li, gi, di)]
            key = 'l%d_g%d_gate_kernel' % (li, gi)
            if key in g:
                base = 'PLE/ple_gate_%d/task_%s/dense/' % (li, tn)
                vals[base + 'kernel'] = g[key]
                vals[base + 'bias'] = g['l%d_g%d_gate_bias' % (li, gi)]
    layer.set_weights_by_name(vals)


def test_ple_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_ple_layer.py:18-38
    from rec_now_amd.layers.ple_layer import PLELayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('ple')
    inputs = torch.from_numpy(g['inputs']).to(dev)
    ple_layer = PLELayer(2, [[2, 3], [2, 3], [3, 2]], [4, 3, 2], 1, name="PLE")
    ple_layer(inputs)
    _load_ple(ple_layer, g)
    task1_output, task2_output = ple_layer(inputs)
    assert calc_sum_of_abs_diff(task1_output, g['golden_task1']) < 1e-5
    assert calc_sum_of_abs_diff(task2_output, g['golden_task2']) < 1e-5


def test_ple_fwd_bwd_vs_oracle(dev, golden):
    from rec_now_amd.layers.ple_layer import PLELayer
    g = golden('ple')
    rng = np.random.default_rng(3)
    B = 333
    x = rng.normal(0, 1, (B, 4)).astype(np.float32)
    layer = PLELayer(2, [[2, 3], [2, 3], [3, 2]], [4, 3, 2], 1, name="PLE", activation='tanh')
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    layer(xd)
    _load_ple(layer, g)
    outs = layer(xd)
    gy = [rng.normal(size=(B, 2)).astype(np.float32) for _ in range(2)]
    (outs[0] * torch.from_numpy(gy[0]).to(dev)).sum().add((outs[1] * torch.from_numpy(gy[1]).to(dev)).sum()).backward()
    g64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in g.items() if k.startswith('l')}
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    routs = R.ple_layer(x64, ple_layers_from_fixture(g64, to=lambda v: v), [True, False, False], activation='tanh')
    ((routs[0] * torch.from_numpy(gy[0]).double()).sum() + (routs[1] * torch.from_numpy(gy[1]).double()).sum()).backward()
    close(outs[0], routs[0])
    close(outs[1], routs[1])
    close(xd.grad, x64.grad)
    tn = _ple_names(0, 1)
    p = layer.named_weights()['PLE/ple_layer_0/task_%s/PLE/ple_layer_0/task_%s/MultiDenseLayer_0/kernel' % (tn, tn)]
    close(p.grad, g64['l0_g1_dnn0_kernel'].grad)
    p = layer.named_weights()['PLE/ple_gate_1/task_shared_0/dense/kernel']
    close(p.grad, g64['l1_g0_gate_kernel'].grad)


def test_ple_param_errors():
    from rec_now_amd.layers.ple_layer import PLELayer
    with pytest.raises(TypeError):
        PLELayer(2, 3, 2)                      # ple_layer.py:42-43
    with pytest.raises(ValueError):
        PLELayer._extend_int_list([], 3)       # ple_layer.py:73-74
    with pytest.raises(TypeError):
        PLELayer._extend_int_list('a', 3)      # ple_layer.py:66-68


# ---- DCN-v1 ------------------------------------------------------------------------------------------------------
def test_dcn_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_dcn_layer.py:18-30
    from rec_now_amd.layers.dcn_layer import DCNLayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('dcn')
    inputs = torch.from_numpy(g['inputs']).to(dev)
    dcn_layer = DCNLayer(3)
    dcn_layer(inputs)
    dcn_layer.set_weights_by_name({k: v for k, v in g.items() if k.startswith(('kernel_', 'bias_'))})
    result = dcn_layer(inputs)
    assert calc_sum_of_abs_diff(result, g['golden']) < 1e-5


@pytest.mark.parametrize('B,D,L,act,use_bias', [(5, 3, 3, None, True), (1000, 64, 2, 'tanh', True), (777, 1024, 3, None, True),
                                                (300, 2048, 4, 'relu', False), (129, 130, 1, 'sigmoid', True),
                                                (64, 1000, 3, None, True),
                                                # beyond the fused register kernels (reference dcn_layer.py:24-31,91-103 has no limit on
                                                # degree_of_cross or input_dim): general streaming path
                                                (600, 1024, 6, 'tanh', True), (300, 8192, 6, None, True), (257, 8200, 2, 'tanh', False),
                                                (100, 1030, 5, 'sigmoid', True), (33, 4100, 1, 'relu', True), (2100, 64, 9, 'tanh', True),
                                                # user callables as activation (keras.activations.get): per-layer route
                                                (300, 96, 3, 'softsign', True), (128, 5000, 2, 'gelu', False)])
def test_dcn_fwd_bwd_vs_oracle(dev, B, D, L, act, use_bias):
    act = {'softsign': torch.nn.functional.softsign, 'gelu': torch.nn.functional.gelu}.get(act, act)
    from rec_now_amd.layers.dcn_layer import DCNLayer
    rng = np.random.default_rng(B + D + L)
    x = rng.normal(0, 1, (B, D)).astype(np.float32)
    ks = [rng.uniform(-1, 1, (D, 1)).astype(np.float32) / np.sqrt(D) for _ in range(L)]
    bs = [rng.uniform(-0.5, 0.5, (1, D)).astype(np.float32) for _ in range(L)]
    gy = rng.normal(size=(B, D)).astype(np.float32)
    layer = DCNLayer(L, activation=act, use_bias=use_bias)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    layer(xd)
    vals = {'kernel_%d' % i: ks[i] for i in range(L)}
    if use_bias:
        vals.update({'bias_%d' % i: bs[i] for i in range(L)})
    layer.set_weights_by_name(vals)
    y = layer(xd)
    y.backward(torch.from_numpy(gy).to(dev))
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    k64 = [torch.from_numpy(k).double().requires_grad_(True) for k in ks]
    b64 = [torch.from_numpy(b).double().requires_grad_(True) for b in bs] if use_bias else None
    ry = R.dcn_layer(x64, k64, b64, act)
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry)
    close(xd.grad, x64.grad)
    for i in range(L):
        close(layer.kernels[i].grad, k64[i].grad)
        if use_bias:
            close(layer.biases[i].grad, b64[i].grad)


# ---- DCN-v2 mix -----------------------------------------------------------------------------------------------------
def _mix_weights(g, L):
    vals = {}
    for l in range(L):
        for n in ('origin_to_sub_kernels_of_layer%d', 'sub_to_sub_kernels_of_layer%d', 'sub_to_origin_kernels_of_layer%d',
                  'bias_of_layer%d'):
            vals[n % l] = g[n % l]
        vals['gate_of_layer%d/kernel' % l] = g['gate_of_layer%d' % l]
    return vals


def test_dcn_mix_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_dcn_mix_layer.py:18-31
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('dcn_mix')
    input = torch.from_numpy(g['inputs']).to(dev)
    dcn_mix_layer = DCNMixLayer(dim_sub_space=3, num_layer=2, num_expert=4)
    dcn_mix_layer(input)
    dcn_mix_layer.set_weights_by_name(_mix_weights(g, 2))
    dcn_mix_output = dcn_mix_layer(input)
    assert calc_sum_of_abs_diff(dcn_mix_output, g['golden']) < 1e-5


@pytest.mark.parametrize('B,D,S,N,L,ai,ao', [(9, 5, 3, 4, 2, 'tanh', 'tanh'), (500, 64, 16, 2, 3, 'tanh', 'tanh'),
                                              (1000, 256, 64, 2, 2, 'relu', 'sigmoid'), (300, 1024, 64, 2, 3, 'tanh', 'tanh'),
                                              (257, 130, 7, 3, 1, None, 'tanh'),
                                              # user callables (reference dcn_mix_layer.py:48-49: keras.activations.get): unfused route
                                              (400, 96, 8, 3, 2, 'softsign', 'tanh'), (300, 256, 64, 2, 2, 'gelu', 'softsign')])
def test_dcn_mix_fwd_bwd_vs_oracle(dev, B, D, S, N, L, ai, ao):
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    named = {'softsign': torch.nn.functional.softsign, 'gelu': torch.nn.functional.gelu}
    ai, ao = named.get(ai, ai), named.get(ao, ao)
    rng = np.random.default_rng(B + D + S)
    x = rng.normal(0, 0.5, (B, D)).astype(np.float32)
    gy = rng.normal(size=(B, D)).astype(np.float32)
    layer = DCNMixLayer(S, num_layer=L, num_expert=N, activation_inner=ai, activation_outer=ao)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    layer(xd)
    w = {k: rng.uniform(-1, 1, tuple(v.shape)).astype(np.float32) * (0.3 if 'bias' in k else 1.5 / np.sqrt(v.shape[-2]))
         for k, v in layer.named_weights().items()}
    layer.set_weights_by_name(w)
    y = layer(xd)
    y.backward(torch.from_numpy(gy).to(dev))
    w64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    pick = lambda fmt: [w64[fmt % l] for l in range(L)]     # noqa: E731
    ry = R.dcn_mix_layer(x64, pick('origin_to_sub_kernels_of_layer%d'), pick('sub_to_sub_kernels_of_layer%d'),
                         pick('sub_to_origin_kernels_of_layer%d'), pick('bias_of_layer%d'), pick('gate_of_layer%d/kernel'), ai, ao)
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry)
    close(xd.grad, x64.grad)
    for k, p in layer.named_weights().items():
        close(p.grad, w64[k].grad)


# ---- CIN ---------------------------------------------------------------------------------------------------------------
def test_cin_reference_golden(dev, golden):
    # /root/reference/tests/layers/test_cin_layer.py:18-42
    from rec_now_amd.layers.cin_layer import CINLayer
    from rec_now_amd.util.numpy_tools import calc_sum_of_abs_diff
    g = golden('cin')
    embeddings = [torch.from_numpy(x).to(dev) for x in g['inputs']]
    cin = CINLayer([2, 1], name="CIN")
    cin(embeddings, output_input=True, sum_channel=True)
    cin.set_weights_by_name({'weight_of_layer1': g['weight_of_layer1'], 'weight_of_layer2': g['weight_of_layer2']})
    cin_output = cin(embeddings, output_input=True, sum_channel=True)
    assert calc_sum_of_abs_diff(cin_output, g['golden']) < 1e-5


def test_cin_tensor_input_needs_embedding_dim(dev):
    from rec_now_amd.layers.cin_layer import CINLayer
    with pytest.raises(ValueError):
        CINLayer([4])(torch.zeros(2, 12, device=dev))          # cin_layer.py:52-54


@pytest.mark.parametrize('B,F,D,Hs,oi,sc,as_list', [(3, 4, 2, [3], True, True, True), (50, 10, 3, [2, 1], True, True, True),
                                                     (64, 8, 4, [16, 8], False, True, False), (40, 6, 4, [5, 7], True, False, True),
                                                     (33, 5, 8, [12], False, False, False), (128, 16, 8, [32, 32, 16], True, True, True)])
def test_cin_fwd_bwd_vs_oracle(dev, B, F, D, Hs, oi, sc, as_list):
    from rec_now_amd.layers.cin_layer import CINLayer
    rng = np.random.default_rng(B + F + D)
    xs = [rng.normal(0, 0.5, (B, D)).astype(np.float32) for _ in range(F)]
    layer = CINLayer(Hs, embedding_dim=D)
    xd = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in xs]
    inp = xd if as_list else torch.cat(xd, dim=1)
    layer(inp, oi, sc)
    ext = [F] + Hs
    w = {'weight_of_layer%d' % k: rng.uniform(-0.5, 0.5, (1, 1, ext[k], ext[k - 1] * F)).astype(np.float32) for k in range(1, len(ext))}
    layer.set_weights_by_name(w)
    y = layer(inp, oi, sc)
    gy = rng.normal(size=tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(gy).to(dev))
    x64 = [torch.from_numpy(x).double().requires_grad_(True) for x in xs]
    w64 = [torch.from_numpy(w['weight_of_layer%d' % k]).double().requires_grad_(True) for k in range(1, len(ext))]
    ry = R.cin_layer(x64, w64, F, D, oi, sc)
    ry.backward(torch.from_numpy(gy).double())
    close(y, ry)
    for a, b in zip(xd, x64):
        close(a.grad, b.grad)
    for k in range(1, len(ext)):
        close(layer.idx2weight[k].grad, w64[k - 1].grad)


# ---- GEMM side product / rank-R epilogue update (the exact-128 formulation of DCN-v2) ------------------------------------
@pytest.mark.parametrize('ta,splitk_shape', [(0, False), (1, True)])
def test_gemm_side_product_and_rank_update(dev, ta, splitk_shape):
    from rec_now_amd import _lib
    rng = np.random.default_rng(17 + ta)
    M, N, K = (256, 128, 8192) if splitk_shape else (512, 128, 256)
    A = rng.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32)
    Bm = rng.uniform(-1, 1, (K, N)).astype(np.float32)
    Bx = rng.uniform(-1, 1, (K, 2)).astype(np.float32)
    Ad, Bd, Bxd = (torch.from_numpy(v).to(dev) for v in (A, Bm, Bx))
    C = torch.empty((M, N), device=dev)
    Cx = torch.full((M, 2), 7.0, device=dev)
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_trans = Ad.data_ptr(), A.shape[1], ta
    d.B, d.ldb = Bd.data_ptr(), N
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = Bxd.data_ptr(), Cx.data_ptr(), 2, 1, 2, 1, 2
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    A64 = A.astype(np.float64).T if ta else A.astype(np.float64)
    bound = 3e-7 * K
    assert np.abs(C.cpu().numpy() - A64 @ Bm.astype(np.float64)).max() <= bound
    assert np.abs(Cx.cpu().numpy() - A64 @ Bx.astype(np.float64)).max() <= bound
    if not ta:
        # rank-2 epilogue update: C = (A B + P Q) * E
        P = rng.uniform(-1, 1, (M, 2)).astype(np.float32)
        Q = rng.uniform(-1, 1, (2, N)).astype(np.float32)
        E = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        Pd, Qd, Ed = (torch.from_numpy(v).to(dev) for v in (P, Q, E))
        d2 = _lib.GemmDesc()
        d2.A, d2.lda = Ad.data_ptr(), K
        d2.B, d2.ldb = Bd.data_ptr(), N
        d2.C, d2.ldc = C.data_ptr(), N
        d2.M, d2.N, d2.K, d2.batch = M, N, K, 1
        d2.emul, d2.lde, d2.e_mode = Ed.data_ptr(), N, 1
        d2.eu_p, d2.eu_q, d2.eu_pms, d2.eu_qrs, d2.eu_qns, d2.eu_r = Pd.data_ptr(), Qd.data_ptr(), 2, N, 1, 2
        _lib.call('recnow_gemm', ctypes.byref(d2), _lib.ptr(ws), ws.numel(), _lib.stream())
        ref = (A64 @ Bm.astype(np.float64) + P.astype(np.float64) @ Q.astype(np.float64)) * E
        assert np.abs(C.cpu().numpy() - ref).max() <= bound


@pytest.mark.parametrize('B,D,S,N,L', [(512, 256, 64, 2, 2), (256, 128, 32, 4, 1), (1024, 1024, 64, 2, 3), (256, 2048, 64, 2, 1)])
def test_dcn_mix_exact128_path_vs_oracle(dev, B, D, S, N, L):
    """Shapes with N*S, D multiples of 128 and B a multiple of 256 take the side-product formulation (dcnmix.hip)."""
    test_dcn_mix_fwd_bwd_vs_oracle(dev, B, D, S, N, L, 'tanh', 'tanh')


def test_dcn_mix_backward_on_two_streams_matches_single_stream(dev, monkeypatch):
    """recnow_dcn_mix_bwd(stream2=...) orders the side stream by events only: gradients are bit-identical."""
    from rec_now_amd.layers import _ops
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    torch.manual_seed(5)
    x = (torch.randn(1024, 256, device=dev) * 0.3)
    layer = DCNMixLayer(dim_sub_space=64, num_layer=3, num_expert=2)
    layer(x[:8])
    grads = []
    for two in (False, True):
        monkeypatch.setattr(_ops, 'DCN_MIX_TWO_STREAMS', two)
        xi = x.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        layer(xi).square().sum().backward()
        torch.cuda.synchronize()
        grads.append([xi.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize('B,D,S,N,L', [(1024, 256, 64, 2, 3), (512, 1024, 64, 2, 1), (300, 130, 7, 3, 2), (257, 64, 16, 2, 1)])
def test_dcn_mix_input_without_gradient(dev, B, D, S, N, L):
    """x is data (requires_grad False, the tf.GradientTape.gradient(loss, weights) case): the forward keeps nothing that only
    dx needs and the backward launches none of the dx products.  The output is bit-identical to the run that also produces
    dx; the weight gradients agree to rounding (a product that no longer carries a dx side output may be split over K, which
    changes the summation order only).  Exact-128 shapes and general shapes, one layer and several."""
    from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
    torch.manual_seed(B + D)
    x = torch.randn(B, D, device=dev) * 0.3
    gy = torch.randn(B, D, device=dev)
    layer = DCNMixLayer(dim_sub_space=S, num_layer=L, num_expert=N)
    layer(x[:8])
    runs = []
    for need in (True, False):
        xi = x.clone().requires_grad_(need)
        for p in layer.parameters():
            p.grad = None
        y = layer(xi)
        y.backward(gy)
        torch.cuda.synchronize()
        assert (xi.grad is not None) == need
        runs.append([y.detach().clone()] + [p.grad.clone() for p in layer.parameters()])
    assert torch.equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())


# ---- persistent short-K kernel: every epilogue form, tile counts above and below one resident wave of workgroups -------------
# K = 144 = nine k-tiles runs the ring schedule (operands two k-tiles ahead, carried across tiles): 8576 x 1024 = 536 tiles is a ragged
# second round on 512 resident workgroups without the XCD-aware order (67 row tiles), 9216 x 1024 = 576 tiles the same with it
@pytest.mark.parametrize('M,N,K,tb', [(256, 256, 144, 0), (1024, 1024, 48, 1), (16384, 1024, 144, 0), (384, 128, 16, 1),
                                      (8576, 1024, 144, 0), (9216, 1024, 144, 1), (8576, 1024, 144, 1)])
@pytest.mark.parametrize('form', ['plain', 'emul', 'accum', 'emul_accum', 'dual_raw', 'dual_fma'])
def test_gemm_short_k_persistent_kernel(dev, M, N, K, tb, form):
    from rec_now_amd import _lib
    if form == 'dual_raw' and tb:
        pytest.skip('C2 = acc is instantiated for B stored [K][N] (DCN-v2 GEMM3)')
    if form == 'dual_fma' and not tb:
        pytest.skip('C2 += acc * E2 is instantiated for B stored [N][K] (DCN-v2 dxl)')
    rng = np.random.default_rng(M + K + tb)
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    Bm = rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32)
    E = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    E2 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    D0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    Ad, Bd, Ed, E2d = (torch.from_numpy(v).to(dev) for v in (A, Bm, E, E2))
    C = torch.from_numpy(C0.copy()).to(dev)
    C2 = torch.from_numpy(D0.copy()).to(dev)
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.A, d.lda = Ad.data_ptr(), K
    d.B, d.ldb, d.b_trans = Bd.data_ptr(), Bm.shape[1], tb
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    P = A.astype(np.float64) @ (Bm.astype(np.float64).T if tb else Bm.astype(np.float64))
    ref, ref2 = P, None
    if form in ('emul', 'emul_accum', 'dual_raw'):
        d.emul, d.lde, d.e_mode = Ed.data_ptr(), N, 1
        ref = ref * E
    if form in ('accum', 'emul_accum'):
        d.accumulate = 1
        ref = ref + C0
    if form == 'dual_raw':
        d.C2, d.ldc2, d.c2_mode = C2.data_ptr(), N, 1
        ref2 = P
    if form == 'dual_fma':
        d.C2, d.ldc2, d.E2, d.lde2, d.c2_mode = C2.data_ptr(), N, E2d.data_ptr(), N, 2
        ref2 = D0 + P * E2
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    bound = 3e-7 * K + 1e-6
    assert np.abs(C.cpu().numpy() - ref).max() <= bound
    if ref2 is not None:
        assert np.abs(C2.cpu().numpy() - ref2).max() <= bound


def test_gemm_second_output_outside_short_k_is_unsupported(dev):
    from rec_now_amd import _lib
    lib = _lib.load()
    A = torch.zeros(128, 1024, device=dev)
    Bm = torch.zeros(1024, 128, device=dev)
    C = torch.zeros(128, 128, device=dev)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = A.data_ptr(), 1024, Bm.data_ptr(), 128, C.data_ptr(), 128
    d.M, d.N, d.K, d.batch = 128, 128, 1024, 1
    d.C2, d.ldc2, d.c2_mode = C.data_ptr(), 128, 1
    ws = _lib.workspace(256, dev)
    assert lib.recnow_gemm(ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream()) == -3


def test_gemm_a_stream_side_output(dev):
    """C = (A*A2) B^T with a side product, while as_out = A * as_in is written from the A tiles the kernel loads anyway."""
    from rec_now_amd import _lib
    rng = np.random.default_rng(23)
    M, N, K = 512, 128, 256
    A, A2, T = (rng.uniform(-1, 1, (M, K)).astype(np.float32) for _ in range(3))
    Bm = rng.uniform(-1, 1, (N, K)).astype(np.float32)
    Bx = rng.uniform(-1, 1, (2, K)).astype(np.float32)
    Ad, A2d, Td, Bd, Bxd = (torch.from_numpy(v).to(dev) for v in (A, A2, T, Bm, Bx))
    C = torch.empty((M, N), device=dev)
    Cx = torch.empty((M, 2), device=dev)
    out = torch.full((M, K), 9.0, device=dev)
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.A, d.A2, d.a_mode, d.lda = Ad.data_ptr(), A2d.data_ptr(), 1, K
    d.B, d.ldb, d.b_trans = Bd.data_ptr(), K, 1
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = Bxd.data_ptr(), Cx.data_ptr(), 1, K, 2, 1, 2
    d.as_in, d.as_out = Td.data_ptr(), out.data_ptr()
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)) + 256, dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    X = (A * A2).astype(np.float64)
    assert np.abs(C.cpu().numpy() - X @ Bm.astype(np.float64).T).max() <= 3e-7 * K
    assert np.abs(Cx.cpu().numpy() - X @ Bx.astype(np.float64).T).max() <= 3e-7 * K
    assert np.array_equal(out.cpu().numpy(), A * T)


# ---- DCN-v2 shape sweep: every dispatch corner (exact-128 / padded path, fused sub-space stage or batched GEMMs, split-K or
# not, tails in B / D / S) against the oracle ---------------------------------------------------------------------------
@pytest.mark.parametrize('B,D,S,N,L', [
    (256, 1024, 64, 2, 2), (768, 1024, 64, 2, 1), (4096, 1024, 64, 2, 1), (8192, 512, 64, 2, 2),      # exact path, B < 32768
    (1000, 1024, 64, 2, 2), (255, 1024, 64, 2, 1),                                                  # B tail -> padded path
    (512, 1000, 64, 2, 1), (512, 96, 64, 2, 2),                                                     # D tail
    (512, 256, 32, 4, 2), (512, 256, 32, 2, 1), (512, 256, 64, 1, 2), (512, 256, 64, 4, 1),         # N*S = 128, 64, 64, 256
    (512, 256, 8, 2, 1), (300, 70, 5, 3, 2), (512, 256, 128, 1, 1), (640, 384, 32, 8, 1)])          # S without a fused kernel
def test_dcn_mix_shape_sweep(dev, B, D, S, N, L):
    test_dcn_mix_fwd_bwd_vs_oracle(dev, B, D, S, N, L, 'tanh', 'tanh')


# ---- GEMM fuzz: random shapes x operand modes x epilogues through every dispatch branch (lean / general / short-K / split-K) ----
def _act_np(v, act):
    return [v, np.maximum(v, 0), np.tanh(v), 1 / (1 + np.exp(-v))][act]


def _actgrad_np(y, act):
    return [np.ones_like(y), (y > 0).astype(np.float64), 1 - y * y, y * (1 - y)][act]


@pytest.mark.parametrize('seed', range(48))
def test_gemm_fuzz(dev, seed):
    from rec_now_amd import _lib
    rng = np.random.default_rng(1000 + seed)
    pick = lambda *v: v[rng.integers(len(v))]                       # noqa: E731
    M = int(pick(1, 37, 128, 256, 384, 512, 1000, 2048))
    N = int(pick(1, 6, 32, 64, 100, 128, 160, 256, 300))
    K = int(pick(1, 16, 48, 144, 256, 288, 1000, 1024, 4096))
    batch = int(pick(1, 1, 1, 2, 3))
    ta, tb = int(pick(0, 1)), int(pick(0, 1))
    a_mode, b_mode = int(pick(0, 0, 1, 2)), int(pick(0, 0, 0, 2))
    act = int(pick(0, 0, 1, 2, 3))
    use_bias, use_emul, accum, c_trans = bool(pick(0, 1)), int(pick(0, 0, 1, 2)), bool(pick(0, 0, 1)), bool(pick(0, 0, 0, 1))
    if c_trans and batch > 1:
        c_trans = False
    A = rng.uniform(-1, 1, (batch, K, M) if ta else (batch, M, K)).astype(np.float32)
    A2 = rng.uniform(0.1, 0.9, A.shape).astype(np.float32)
    Bm = rng.uniform(-1, 1, (batch, N, K) if tb else (batch, K, N)).astype(np.float32)
    B2 = rng.uniform(0.1, 0.9, Bm.shape).astype(np.float32)
    bias = rng.uniform(-1, 1, (batch, N)).astype(np.float32)
    E = rng.uniform(0.1, 0.9, (batch, M, N)).astype(np.float32)
    C0 = rng.uniform(-1, 1, (batch, N, M) if c_trans else (batch, M, N)).astype(np.float32)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)     # noqa: E731
    Ad, A2d, Bd, B2d, bd, Ed, Cd = t(A), t(A2), t(Bm), t(B2), t(bias), t(E), t(C0.copy())
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_batch_stride, d.a_trans = Ad.data_ptr(), A.shape[2], A.shape[1] * A.shape[2], ta
    d.B, d.ldb, d.b_batch_stride, d.b_trans = Bd.data_ptr(), Bm.shape[2], Bm.shape[1] * Bm.shape[2], tb
    d.C, d.ldc, d.c_batch_stride, d.c_trans = Cd.data_ptr(), C0.shape[2], C0.shape[1] * C0.shape[2], int(c_trans)
    d.M, d.N, d.K, d.batch = M, N, K, batch
    a_act, b_act, e_act = int(pick(1, 2, 3)), int(pick(1, 2, 3)), int(pick(1, 2, 3))
    A64, B64 = A.astype(np.float64), Bm.astype(np.float64)
    if a_mode:
        d.A2, d.a_mode, d.a_act = A2d.data_ptr(), a_mode, a_act
        A64 = A64 * (A2 if a_mode == 1 else _actgrad_np(A2.astype(np.float64), a_act))
    if b_mode:
        d.B2, d.b_mode, d.b_act = B2d.data_ptr(), b_mode, b_act
        B64 = B64 * _actgrad_np(B2.astype(np.float64), b_act)
    if ta:
        A64 = A64.transpose(0, 2, 1)
    if tb:
        B64 = B64.transpose(0, 2, 1)
    ref = A64 @ B64
    mag = np.abs(A64) @ np.abs(B64)
    if use_bias:
        d.bias, d.bias_batch_stride = bd.data_ptr(), N
        ref = ref + bias[:, None, :]
    d.act = act
    ref = _act_np(ref, act)
    if use_emul:
        d.emul, d.lde, d.e_batch_stride, d.e_mode, d.e_act = Ed.data_ptr(), N, M * N, use_emul, e_act
        ref = ref * (E if use_emul == 1 else _actgrad_np(E.astype(np.float64), e_act))
    if accum:
        d.accumulate = 1
        ref = ref + (C0.transpose(0, 2, 1) if c_trans else C0)
    lib = _lib.load()
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)) + 256, dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    out = Cd.cpu().numpy().astype(np.float64)
    if c_trans:
        out = out.transpose(0, 2, 1)
    bound = 3e-7 * mag.max() + 2e-6
    assert np.abs(out - ref).max() <= bound, (M, N, K, batch, ta, tb, a_mode, b_mode, act, use_bias, use_emul, accum, c_trans)


# ---- layer shape sweeps at sizes that reach the lean / split-K / head kernels (the small cases above mostly run the general
# kernels): MultiDense with 1..8 and hundreds of output columns, CIN with tile-aligned and ragged widths ---------------------
@pytest.mark.parametrize('B,D,U,N,batched,act', [
    (4096, 1024, 1, 1, False, None), (4096, 1024, 1, 1, False, 'sigmoid'),                      # scoring head kernels
    (4096, 1024, 6, 1, False, None), (2048, 4096, 8, 1, False, 'tanh'),                         # narrow outputs -> MFMA GEMM
    (2048, 1024, 4, 2, False, 'tanh'), (1024, 512, 256, 2, False, 'tanh'),
    (1024, 512, 256, 2, True, 'tanh'), (8192, 256, 128, 1, False, 'sigmoid'),
    (1000, 333, 77, 3, True, 'tanh'), (256, 2048, 512, 4, False, None),
    # narrow layers with a long batch (SENET's excitation MLP): register-tile weight gradient instead of the split-K GEMM
    (5000, 64, 32, 1, False, 'tanh'), (4096, 32, 64, 1, False, None), (9001, 128, 32, 1, False, 'sigmoid'), (4500, 16, 16, 1, False, 'relu')])
def test_multi_dense_shape_sweep(dev, B, D, U, N, batched, act):
    test_multi_dense_fwd_bwd_vs_oracle(dev, B, D, U, N, batched, act)


@pytest.mark.parametrize('B,F,D,Hs,oi,sc,as_list', [
    (512, 16, 16, [32, 32], True, True, True), (256, 64, 16, [128, 128], True, True, True), (256, 8, 32, [64], False, True, False),
    (300, 12, 16, [128, 20, 128], True, False, True), (1024, 4, 8, [256], False, False, True), (128, 33, 4, [65, 31], True, True, True),
    # the fused data-gradient kernel of the backward pass (csrc/cin_bwd.hip: rows a multiple of 128, H_{k-1} 64 or 128): two fields per column tile
    # with separate / shared output buffers, one field per tile, a single k-tile, layers that fall back to the two products in between
    (96, 12, 16, [64, 128, 32], True, True, True), (64, 64, 16, [128, 64, 128], False, False, False), (128, 20, 8, [128, 128], True, True, True)])
def test_cin_shape_sweep(dev, B, F, D, Hs, oi, sc, as_list):
    test_cin_fwd_bwd_vs_oracle(dev, B, F, D, Hs, oi, sc, as_list)


@pytest.mark.parametrize('B,D,L', [(700, 1024, 3), (300, 1024, 7), (200, 6000, 2)])
def test_dcn_backward_without_saved_scalars_matches(dev, B, D, L):
    """recnow_dcn_bwd(csave = NULL) recomputes the forward per row; with csave it uses the forward's scalars: same gradients
    (fused kernels and, for L > 4 or D > 4096, the general path through the C ABI)."""
    from rec_now_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(8)
    x, dy = (torch.from_numpy(rng.normal(0, 1, (B, D)).astype(np.float32)).to(dev) for _ in range(2))
    k = torch.from_numpy((rng.uniform(-1, 1, (L, D)) / np.sqrt(D)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-0.5, 0.5, (L, D)).astype(np.float32)).to(dev)
    y = torch.empty_like(x)
    cs = torch.empty((B, L), device=dev)
    _lib.call('recnow_dcn_fwd', _lib.ptr(x), _lib.ptr(k), _lib.ptr(b), B, D, L, 2, _lib.ptr(y), _lib.ptr(cs), _lib.stream())
    ws = _lib.workspace(lib.recnow_dcn_workspace_bytes(B, D, L), dev)
    outs = []
    for c in (cs, None):
        dx, dk, db = torch.empty_like(x), torch.empty_like(k), torch.empty_like(b)
        _lib.call('recnow_dcn_bwd', _lib.ptr(x), _lib.ptr(k), _lib.ptr(b), _lib.ptr(dy), _lib.ptr(c), B, D, L, 2, _lib.ptr(dx), _lib.ptr(dk),
                  _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.stream())
        outs.append((dx, dk, db))
    for a, r in zip(*outs):
        close(a, r, rtol=2e-6)
