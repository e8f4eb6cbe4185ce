"""Regenerates tests/golden/*.npz.  Run from the repo root:  python tests/golden/make_golden.py

A fixture = the inputs + weights of one of the reference's own hot-path unit tests (regenerated with
oracle/tf_seeded_rng.py, because the tests seed TensorFlow's RNG and TF is not installable here) plus
the literal golden output that test asserts (transcribed DATA, cited by file:line below).  No reference
source is copied.  The script refuses to write a fixture unless the dense oracle reproduces the golden
within the reference's own tolerance (sum|diff| < 1e-5, util/numpy_tools.py:12-27).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import dense_ref as R                      # noqa: E402
from tf_seeded_rng import TFSeededRNG      # noqa: E402

T = torch.from_numpy
TOL = 1e-5


def _save(name, golden, got, **arrays):
    diff = R.calc_sum_of_abs_diff(got, golden)
    assert diff < TOL, '%s: oracle does not reproduce the reference golden (sum|diff| = %g)' % (name, diff)
    np.savez(os.path.join(HERE, name + '.npz'), golden=np.asarray(golden, np.float32), **arrays)
    print('%-16s sum|oracle-golden| = %.3g' % (name, diff))


def fm():
    # tests/layers/test_fm_layer.py:19-35
    r = TFSeededRNG(1)
    xs = [r.uniform([2, 5], 0.0, 1.0, seed=f) for f in range(7)]
    golden = [[17.120375], [32.70206]]
    got = R.fm_layer([T(x) for x in xs]).numpy()
    _save('fm', golden, got, inputs=np.stack(xs))


def dcn():
    # tests/layers/test_dcn_layer.py:19-28
    r = TFSeededRNG(1)
    x = np.array([[1, 2, 3], [4, 5, 6]], np.float32)
    ks = [r.glorot_uniform([3, 1]) for _ in range(3)]
    bs = [np.zeros((1, 3), np.float32) for _ in range(3)]
    golden = [[-3.0642402, -6.1284804, -9.19272], [-140.33298, -175.41623, -210.49947]]
    got = R.dcn_layer(T(x), [T(k) for k in ks], [T(b) for b in bs]).numpy()
    _save('dcn', golden, got, inputs=x, **{'kernel_%d' % i: k for i, k in enumerate(ks)},
          **{'bias_%d' % i: b for i, b in enumerate(bs)})


def multi_dense():
    # tests/layers/test_multi_dense_layer.py:19-36 (2-D input) and :38-55 (3-D input)
    for name, shape, golden in (
            ('multi_dense_2d', [2, 4], [[[-0.03584324], [-0.004118]], [[-0.03538369], [-0.00670193]],
                                        [[0.0174423], [-0.01468011]]]),
            ('multi_dense_3d', [3, 2, 4], [[[-0.03584324], [-0.004118]], [[-0.0016742], [0.00547126]],
                                           [[0.00360582], [-0.03530192]]])):
        r = TFSeededRNG(1)
        x = r.random_normal_initializer(shape)
        k = r.glorot_uniform([3, 4, 1])
        b = np.zeros((3, 1, 1), np.float32)
        got = R.multi_dense_layer(T(x), T(k), T(b)).numpy()
        _save(name, golden, got, inputs=x, kernel=k, bias=b)


def mmoe():
    # tests/layers/test_mmoe_layer.py:19-36
    r = TFSeededRNG(1)
    x = np.array([[1, 2, 3, 4], [4, 5, 6, 7], [4, 5, 6, 7]], np.float32)
    ek = [r.glorot_uniform([4, 4, 8]), r.glorot_uniform([4, 8, 3])]
    eb = [np.zeros((4, 1, 8), np.float32), np.zeros((4, 1, 3), np.float32)]
    gk = r.glorot_uniform([2, 4, 4])
    gb = np.zeros((2, 1, 4), np.float32)
    golden = [[[0.14612462, 0.44929513, -0.78639925], [0.4357101, 1.235869, -1.852003],
               [0.4357101, 1.235869, -1.852003]],
              [[0.3218293, -0.01983448, -0.7208969], [0.8113805, -0.34114015, -1.272174],
               [0.8113805, -0.34114015, -1.272174]]]
    got = R.mmoe_layer(T(x), [T(k) for k in ek], [T(b) for b in eb], T(gk), T(gb)).numpy()
    _save('mmoe', golden, got, inputs=x, expert_kernel_0=ek[0], expert_kernel_1=ek[1],
          expert_bias_0=eb[0], expert_bias_1=eb[1], gate_kernel=gk, gate_bias=gb)


def dcn_mix():
    # tests/layers/test_dcn_mix_layer.py:21-29
    r = TFSeededRNG(10)
    x = np.array([[1, 2, 3, 4, 5], [10, 11, 12, 13, 14]], np.float32)
    L, N, D, S = 2, 4, 5, 3
    U = [r.glorot_uniform([N, D, S]) for _ in range(L)]
    V = [r.glorot_uniform([N, S, S]) for _ in range(L)]
    W = [r.glorot_uniform([N, S, D]) for _ in range(L)]
    b = [np.zeros((1, N, D), np.float32) for _ in range(L)]
    K = [r.glorot_uniform([D, N]) for _ in range(L)]          # gate Dense kernels, created at call time
    golden = [[-0.00718718, -0.05909997, 0.04065184, 0.06140723, -0.05879733],
              [-0.35002837, -1.4658885, 1.1511558, 1.3849638, -1.1614282]]
    tl = lambda a: [T(v) for v in a]                          # noqa: E731
    got = R.dcn_mix_layer(T(x), tl(U), tl(V), tl(W), tl(b), tl(K)).numpy()
    arrays = {}
    for l in range(L):
        arrays['origin_to_sub_kernels_of_layer%d' % l] = U[l]
        arrays['sub_to_sub_kernels_of_layer%d' % l] = V[l]
        arrays['sub_to_origin_kernels_of_layer%d' % l] = W[l]
        arrays['bias_of_layer%d' % l] = b[l]
        arrays['gate_of_layer%d' % l] = K[l]
    _save('dcn_mix', golden, got, inputs=x, **arrays)


def cin():
    # tests/layers/test_cin_layer.py:19-40
    r = TFSeededRNG(1)
    tables = [r.random_normal_initializer([5, 3]) for _ in range(10)]
    embs = [t[[0, 1]] for t in tables]                        # features = [0, 1] for every field
    w1 = r.glorot_uniform([1, 1, 2, 20 // 2 * 10])            # [1,1,H1=2,H0*F=10*10]
    w2 = r.glorot_uniform([1, 1, 1, 2 * 10])                  # [1,1,H2=1,H1*F=2*10]
    golden = [[0.00754739, 0.15105832, 0.23957989], [0.03649065, -0.12252036, -0.02113211]]
    got = R.cin_layer([T(e) for e in embs], [T(w1), T(w2)], 10, 3).numpy()
    _save('cin', golden, got, inputs=np.stack(embs), weight_of_layer1=w1, weight_of_layer2=w2)


def ple():
    # tests/layers/test_ple_layer.py:19-36
    r = TFSeededRNG(1)
    x = np.array([[1, 2, 3, 4], [4, 5, 6, 7]], np.float32)
    dnn_dims = [[2, 3], [2, 3], [3, 2]]
    n_exp = [4, 3, 2]
    is_shared = [True, False, False]
    arrays = {}
    layers = []
    din_per_group = [4, 4, 4]
    for li in range(3):
        is_last = li == 2
        layer = {'dnn': [], 'gate': []}
        for gi in range(3):                                    # experts, in call order
            din = din_per_group[gi]
            dnn = []
            for di, u in enumerate(dnn_dims[li]):
                k = r.glorot_uniform([n_exp[li], din, u])
                b = np.zeros((n_exp[li], 1, u), np.float32)
                arrays['l%d_g%d_dnn%d_kernel' % (li, gi, di)] = k
                arrays['l%d_g%d_dnn%d_bias' % (li, gi, di)] = b
                dnn.append((T(k), T(b)))
                din = u
            layer['dnn'].append(dnn)
        for gi in range(3):                                    # then gates, in call order
            if is_shared[gi] and is_last:
                layer['gate'].append(None)
                continue
            units = 3 * n_exp[li] if is_shared[gi] else 2 * n_exp[li]
            k = r.glorot_uniform([din_per_group[gi], units])
            b = np.zeros((units,), np.float32)
            arrays['l%d_g%d_gate_kernel' % (li, gi)] = k
            arrays['l%d_g%d_gate_bias' % (li, gi)] = b
            layer['gate'].append((T(k), T(b)))
        layers.append(layer)
        u_out = dnn_dims[li][-1]
        din_per_group = [3 * u_out, 2 * u_out, 2 * u_out]
    got = R.ple_layer(T(x), layers, is_shared)
    golden1 = [[-0.00202228, 0.03448214], [0.00163524, 0.13148618]]
    golden2 = [[-0.00116823, 0.01959552], [0.00839254, 0.04192837]]
    d1 = R.calc_sum_of_abs_diff(got[0].numpy(), golden1)
    d2 = R.calc_sum_of_abs_diff(got[1].numpy(), golden2)
    assert d1 < TOL and d2 < TOL, (d1, d2)
    np.savez(os.path.join(HERE, 'ple.npz'), inputs=x, golden_task1=np.asarray(golden1, np.float32),
             golden_task2=np.asarray(golden2, np.float32), **arrays)
    print('%-16s sum|oracle-golden| = %.3g / %.3g' % ('ple', d1, d2))


def inner_pnn():
    # tests/layers/test_inner_pnn_layer.py:18-39
    r = TFSeededRNG(1)
    xs = [r.normal([4, 2], mean=float(f), stddev=1.0, seed=f) for f in range(3)]
    golden = [[2.3125873, 1.9014311, 10.842302], [1.5337583, 2.2996788, 13.424303],
              [-0.799806, -8.9576435, 1.9083395], [0.07315224, 1.5779386, 4.7213144]]
    got = R.inner_pnn_layer([T(x) for x in xs]).numpy()
    _save('inner_pnn', golden, got, inputs=np.stack(xs))


def senet():
    # tests/layers/test_senet_layer.py:18-39: fields of width 1, 2, 3; reduction_ratio 0.3 -> middle_dim max(round(0.9), 1) = 1
    r = TFSeededRNG(1)
    xs = [r.random_normal_initializer([2, d]) for d in (1, 2, 3)]
    k0 = r.glorot_uniform([3, 1])
    k1 = r.glorot_uniform([1, 3])
    b0, b1 = np.zeros((1,), np.float32), np.zeros((3,), np.float32)
    golden = [[-0.00147045, -0.00081997, 0.00221329, 0.00113419, 0.00100974, -0.00180815],
              [-0.00311359, -0.00019354, 0.00409976, -0.00334057, 0.00116946, 0.00371958]]
    got = R.senet_layer([T(x) for x in xs], [T(k0), T(k1)], [T(b0), T(b1)]).numpy()
    _save('senet', golden, got, input_0=xs[0], input_1=xs[1], input_2=xs[2], dense_0_kernel=k0, dense_1_kernel=k1,
          dense_0_bias=b0, dense_1_bias=b1)


if __name__ == '__main__':
    fm(); dcn(); multi_dense(); mmoe(); dcn_mix(); cin(); ple(); inner_pnn(); senet()
