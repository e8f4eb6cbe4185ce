"""GPU parity of the row-block persistent forward of DCNMixLayer (csrc/dcnmix_tile.hip: every cross layer + the folded scoring head in
ONE launch, a workgroup per 32-row block; reference /root/reference/rec_now/layers/dcn_mix_layer.py:123-150 -> multi_dense_layer.py:90-92)
against (a) the fp64 oracle and (b) the launch-per-product route it replaces at shard sizes (`RECNOW_TILE=0`; the switch is read per call).
The backward pass consumes what the forward saved (T1, T2, T2g, O_l), so every gradient is part of the comparison."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _chunked_oracle import close
from test_fused_gpu import _build, _oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _route(mode, bwd='1'):
    """mode '1': row-block forward and (bwd '1') the row-block backward chain, bwd '0': the product-route backward behind it; mode '0': the
    launch-per-product route; None: the defaults."""
    if mode is None:
        os.environ.pop('RECNOW_TILE', None)
        os.environ.pop('RECNOW_TILE_BWD', None)
    else:
        os.environ['RECNOW_TILE'] = mode
        os.environ['RECNOW_TILE_BWD'] = bwd


def _run_fused(dev, cross, head, xd, gs):
    from rec_now_amd.fused import dcn_mix_score
    for p in list(cross.parameters()) + list(head.parameters()):
        p.grad = None
    xd.grad = None
    s = dcn_mix_score(cross, head, xd)
    s.backward(torch.from_numpy(gs).to(dev))
    out = {k: p.grad.clone() for k, p in cross.named_weights().items()}
    out['hk'], out['hb'], out['dx'], out['s'] = head.kernel.grad.clone(), head.bias.grad.clone(), xd.grad.clone(), s.detach().clone()
    return out


# 8448 rows = 264 row blocks on 256 workgroups: eight workgroups walk a second block
@pytest.mark.parametrize('B,D,L,ai,ao', [(512, 256, 2, 'tanh', 'tanh'), (1024, 512, 3, 'tanh', 'tanh'), (2048, 1024, 3, 'tanh', 'tanh'),
                                          (8448, 1024, 3, 'tanh', 'tanh'), (768, 1024, 1, 'relu', 'sigmoid'), (1280, 512, 2, None, 'tanh'),
                                          (512, 256, 4, 'tanh', 'tanh')])      # four layers: row-block forward, product-route backward (two g buffers)
def test_tile_forward_vs_oracle_and_product_route(dev, B, D, L, ai, ao):
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, 64, 2, L, B + D + L, ai, ao)
    gs = np.random.default_rng(1).normal(size=B).astype(np.float32)
    try:
        _route('1')
        tile = _run_fused(dev, cross, head, xd, gs)
        _route('1', bwd='0')             # the A/B pairing: row-block forward, product-route backward
        mixed = _run_fused(dev, cross, head, xd, gs)
        _route('0')
        prod = _run_fused(dev, cross, head, xd, gs)
    finally:
        _route(None)
    rs, x64, w64, hk64, hb64 = _oracle(x, w, hk, hb, L, gs, ai, ao)
    close(tile['s'], rs, what='scores')
    close(tile['dx'], x64.grad, what='dx')
    for k in w:
        close(tile[k], w64[k].grad, what=k)
    close(tile['hk'], hk64.grad, what='head kernel')
    close(tile['hb'], hb64.grad, what='head bias', scale=np.abs(gs).sum())
    assert not torch.equal(tile['s'], prod['s'])          # two routes (another summation order), not one route twice
    assert torch.equal(tile['s'], mixed['s'])          # same forward ...
    assert torch.equal(tile['dx'], mixed['dx']) == (L > 3)          # ... another backward (up to three layers: deeper stacks take the product route anyway)
    for k in tile:
        close(tile[k], prod[k], rtol=4e-6, what=k + ' tile vs product route', scale=(np.abs(gs).sum() if k == 'hb' else None))
        close(mixed[k], prod[k], rtol=4e-6, what=k + ' tile forward + product backward vs product route', scale=(np.abs(gs).sum() if k == 'hb' else None))
    close(mixed['dx'], x64.grad, what='dx (tile forward, product backward)')
    for k in w:
        close(mixed[k], w64[k].grad, what=k + ' (tile forward, product backward)')


# 8448 rows: eight workgroups walk a second block (the weight ring of the next block is requested in the last steps of the block before)
@pytest.mark.parametrize('B,D,L,ai,ao', [(512, 256, 2, 'tanh', 'tanh'), (1024, 512, 3, 'tanh', 'tanh'), (8448, 1024, 3, 'tanh', 'tanh'),
                                          (768, 1024, 1, 'relu', 'sigmoid'), (512, 256, 4, None, 'tanh')])
def test_split_precision_tile_forward_vs_oracle_and_split_product_route(dev, B, D, L, ai, ao):
    """Split-precision mode with RECNOW_TILE_SPLIT=1: every cross layer + the folded head in ONE row-block launch on the bf16 MFMA (three pieces, six
    terms: csrc/dcnmix_tile_split.hip), the launch-per-product backward behind it reads what that forward saved -- scores and EVERY gradient against
    the fp64 oracle at the suite's 1e-5 bound, and against the split-precision product route (another summation order, not bit-identical)."""
    from rec_now_amd import _lib
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, 64, 2, L, B + D + L + 1, ai, ao)
    gs = np.random.default_rng(1).normal(size=B).astype(np.float32)
    lib = _lib.load()
    try:
        _lib.call('recnow_set_gemm_precision', 1)
        os.environ['RECNOW_TILE_SPLIT'] = '1'
        assert lib.recnow_dcn_mix_tile_route(B, D, 64, 2, L) == 2
        tile = _run_fused(dev, cross, head, xd, gs)
        os.environ['RECNOW_TILE_SPLIT'] = '0'
        assert lib.recnow_dcn_mix_tile_route(B, D, 64, 2, L) == 0
        prod = _run_fused(dev, cross, head, xd, gs)
    finally:
        os.environ.pop('RECNOW_TILE_SPLIT', None)
        _lib.call('recnow_set_gemm_precision', 0)
    rs, x64, w64, hk64, hb64 = _oracle(x, w, hk, hb, L, gs, ai, ao)
    close(tile['s'], rs, what='scores')
    close(tile['dx'], x64.grad, what='dx')
    for k in w:
        close(tile[k], w64[k].grad, what=k)
    close(tile['hk'], hk64.grad, what='head kernel')
    close(tile['hb'], hb64.grad, what='head bias', scale=np.abs(gs).sum())
    assert not torch.equal(tile['s'], prod['s'])          # two routes, not one route twice
    for k in tile:
        close(tile[k], prod[k], rtol=4e-6, what=k + ' split tile forward vs split product route', scale=(np.abs(gs).sum() if k == 'hb' else None))


def test_tile_forward_of_the_layer_without_head_and_without_input_gradient(dev):
    """`DCNMixLayer.call` alone (recnow_dcn_mix_fwd: the layer output y leaves the last layer) with x as data: O_{L-1} is not kept."""
    B, D, L = 1536, 1024, 3
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, 64, 2, L, 11)
    gy = torch.from_numpy(np.random.default_rng(2).normal(size=(B, D)).astype(np.float32)).to(dev)
    res = {}
    try:
        for mode in ('1', '0'):
            _route(mode)
            for need_dx in (True, False):
                for p in cross.parameters():
                    p.grad = None
                xin = xd.detach().clone().requires_grad_(need_dx)
                y = cross(xin)
                y.backward(gy)
                res[mode, need_dx] = (y.detach().clone(), None if not need_dx else xin.grad.clone(), {k: p.grad.clone() for k, p in cross.named_weights().items()})
    finally:
        _route(None)
    w64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    import dense_ref as R
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    pick = lambda fmt: [w64[fmt % l] for l in range(L)]     # noqa: E731
    y64 = R.dcn_mix_layer(x64, pick('origin_to_sub_kernels_of_layer%d'), pick('sub_to_sub_kernels_of_layer%d'),
                          pick('sub_to_origin_kernels_of_layer%d'), pick('bias_of_layer%d'), pick('gate_of_layer%d/kernel'), 'tanh', 'tanh')
    y64.backward(gy.cpu().double())
    for need_dx in (True, False):
        y, dx, g = res['1', need_dx]
        close(y, y64, what='y')
        if need_dx:
            close(dx, x64.grad, what='dx')
        for k in w:
            close(g[k], w64[k].grad, what=k)
            close(g[k], res['0', need_dx][2][k], rtol=4e-6, what=k + ' tile vs product route')


def test_tile_forward_with_materialised_layer_inputs(dev):
    """RECNOW_XLESS=0 (read once per process: a subprocess): the row-block kernel also writes x_{l+1} = x * O_l for the backward pass."""
    env = dict(os.environ, RECNOW_XLESS='0')
    out = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', os.path.join(ROOT, 'tests', 'test_tile_gpu.py'),
                          '-k', 'vs_oracle_and_product_route and 2048'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout
