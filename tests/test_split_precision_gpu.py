"""GPU parity of the OPT-IN split-precision products (rec_now_amd/csrc/gemm_split.hip: fp32 operands as three bf16 pieces, six bf16
MFMA terms per product, fp32 accumulation; `recnow_set_gemm_precision(1)` / RECNOW_GEMM_PRECISION=bf16x3) against fp64, through
the C ABI: the five (layout, operand kind) products of the DCN-v2 step with their side products, at small sizes and at the step's
own depths (K = 1024 rows x 65 536, K = 65 536 split-K), and the north-star layer end to end.  Same bound as the exact kernels:
1e-5 relative (north_star), written in every assert.  Reference: /root/reference/rec_now/layers/dcn_mix_layer.py:114-151."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def split_mode():
    from rec_now_amd import _lib
    lib = _lib.load()
    assert lib.recnow_get_gemm_precision() == 0
    _lib.call('recnow_set_gemm_precision', 1)
    yield
    _lib.call('recnow_set_gemm_precision', 0)


def _product(dev, M, N, K, ta, tb, mul, seed, scale=1.0):
    """C = op(A [* A2]) op(B), Cx = op(A [* A2]) Bx through recnow_gemm; returns GPU results and the fp64 references."""
    from rec_now_amd import _lib
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((K, M) if ta else (M, K)) * scale).astype(np.float32)
    A2 = rng.uniform(-1, 1, A.shape).astype(np.float32) if mul else None
    Bm = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    Bx = rng.standard_normal((K, 2)).astype(np.float32)
    Ad, Bd, Bxd = (torch.from_numpy(v).to(dev) for v in (A, Bm, Bx))
    A2d = torch.from_numpy(A2).to(dev) if mul else None
    C = torch.empty((M, N), device=dev)
    Cx = torch.full((M, 2), 7.0, device=dev)
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_trans = Ad.data_ptr(), A.shape[1], ta
    if mul:
        d.A2, d.a_mode = A2d.data_ptr(), 1
    d.B, d.ldb, d.b_trans = Bd.data_ptr(), Bm.shape[1], tb
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = Bxd.data_ptr(), Cx.data_ptr(), 2, 1, 2, 1, 2
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    torch.cuda.synchronize()
    A64 = A.astype(np.float64) * (A2.astype(np.float64) if mul else 1.0)
    A64 = A64.T if ta else A64
    B64 = Bm.astype(np.float64).T if tb else Bm.astype(np.float64)
    return C.cpu().numpy(), Cx.cpu().numpy(), A64 @ B64, A64 @ Bx.astype(np.float64)


# (a_trans, b_trans, mul): GEMM1, dT2g, top-layer dT2g under the fused head, dU, dW^T -- the launcher's table in gemm_split.hip
COMBOS = [(0, 0, 0), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1)]


@pytest.mark.parametrize('ta,tb,mul', COMBOS)
@pytest.mark.parametrize('M,K', [(256, 512), (1024, 1024), (256, 8192)])
def test_split_product_vs_fp64(dev, split_mode, ta, tb, mul, M, K):
    C, Cx, R, Rx = _product(dev, M, 128, K, ta, tb, mul, 100 * ta + 10 * tb + mul + K)
    assert np.abs(C - R).max() <= 1e-5 * np.abs(R).max(), (np.abs(C - R).max(), np.abs(R).max())
    assert np.abs(Cx - Rx).max() <= 1e-5 * np.abs(Rx).max()
    # element-wise: the error of one output stays within a few fp32 roundings of its own terms (sum |a||b| ~ K * 0.64)
    assert np.abs(C - R).max() <= 4e-7 * K


def test_split_mode_runs_the_split_kernel_and_differs_from_fp32(dev):
    """The mode switch is real: same inputs, the two modes agree to 1e-5 but are not bit-identical; turning it off restores
    the exact kernel bit for bit."""
    from rec_now_amd import _lib
    C0, _, R, _ = _product(dev, 512, 128, 1024, 0, 0, 0, 5)
    _lib.call('recnow_set_gemm_precision', 1)
    try:
        assert _lib.load().recnow_get_gemm_precision() == 1
        C1, _, _, _ = _product(dev, 512, 128, 1024, 0, 0, 0, 5)
    finally:
        _lib.call('recnow_set_gemm_precision', 0)
    C2, _, _, _ = _product(dev, 512, 128, 1024, 0, 0, 0, 5)
    assert np.array_equal(C0, C2)
    assert not np.array_equal(C0, C1)
    assert np.abs(C1 - R).max() <= 1e-5 * np.abs(R).max()
    assert _lib.load().recnow_set_gemm_precision(7) != 0          # unknown mode: refused


@pytest.mark.parametrize('ta,tb,mul,M,K', [(0, 0, 0, 65536, 1024), (0, 1, 1, 65536, 1024), (1, 0, 0, 1024, 65536), (1, 0, 1, 1024, 65536)])
def test_split_product_at_step_size(dev, split_mode, ta, tb, mul, M, K):
    """The step's own shapes (B = 65 536 rows x D = 1024): forward / dT2g products and the K = B weight-gradient products (split-K
    slabs + deterministic reduce).  Wide dynamic range on purpose (values over 4 decades)."""
    C, Cx, R, Rx = _product(dev, M, 128, K, ta, tb, mul, 7 + ta + mul, scale=0.05)
    assert np.abs(C - R).max() <= 1e-5 * np.abs(R).max(), (np.abs(C - R).max(), np.abs(R).max())
    assert np.abs(Cx - Rx).max() <= 1e-5 * np.abs(Rx).max()


def test_split_pieces_cover_extreme_magnitudes(dev, split_mode):
    """Operands spanning 1e-30 .. 1e30 (bf16 keeps the fp32 exponent range: no scaling step): relative error per output
    against its own magnitude, rows scaled independently."""
    from rec_now_amd import _lib
    rng = np.random.default_rng(3)
    M, N, K = 256, 128, 512
    row_scale = (10.0 ** rng.uniform(-15, 15, (M, 1))).astype(np.float32)
    A = (rng.standard_normal((M, K)).astype(np.float32) * row_scale)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    Bx = rng.standard_normal((K, 2)).astype(np.float32)
    Ad, Bd, Bxd = (torch.from_numpy(v).to(dev) for v in (A, Bm, Bx))
    C = torch.empty((M, N), device=dev)
    Cx = torch.empty((M, 2), device=dev)
    d = _lib.GemmDesc()
    d.A, d.lda = Ad.data_ptr(), K
    d.B, d.ldb = Bd.data_ptr(), N
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = Bxd.data_ptr(), Cx.data_ptr(), 2, 1, 2, 1, 2
    ws = _lib.workspace(_lib.load().recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    R = A.astype(np.float64) @ Bm.astype(np.float64)
    err = np.abs(C.cpu().numpy() - R) / np.abs(R).max(axis=1, keepdims=True)
    assert err.max() <= 1e-5, err.max()


@pytest.mark.parametrize('route', ['layers', 'fused'])
def test_split_mode_dcn_mix_model_vs_oracle(dev, split_mode, route):
    """DCNMixLayer + head under the split products: scores, d/dx and every weight gradient against the fp64 oracle (same
    comparison as tests/test_fused_gpu.py, same 1e-5 bound)."""
    import dense_ref as R
    from _chunked_oracle import close
    from rec_now_amd.fused import dcn_mix_score
    from test_fused_gpu import _build, _oracle
    B, D, S, N, L = 2048, 1024, 64, 2, 3
    x, xd, cross, head, w, hk, hb = _build(dev, B, D, S, N, L, 77)
    gs = np.random.default_rng(1).normal(size=B).astype(np.float32)
    s = dcn_mix_score(cross, head, xd) if route == 'fused' else head(cross(xd)).reshape(-1)
    s.backward(torch.from_numpy(gs).to(dev))
    rs, x64, w64, hk64, hb64 = _oracle(x, w, hk, hb, L, gs)
    close(s, rs, what='scores')
    close(xd.grad, x64.grad, what='dx')
    for k, p in cross.named_weights().items():
        close(p.grad, w64[k].grad, what=k)
    close(head.kernel.grad, hk64.grad, what='head kernel')


@pytest.mark.parametrize('M,K', [(256, 512), (1024, 8192), (1024, 65536)])
def test_lds_dma_staging_is_bit_identical_to_register_staging(dev, M, K):
    """`recnow_set_gemm_staging(1)` (RECNOW_GEMM_GLDS=1) moves both operands of the plain [k][row] products -- dU_l = x_l^T dA with dgate as the side
    product, here at small sizes, split over K, and at the step's own depth -- global -> LDS by LDS-DMA (`k_gemm<.., 41>`, csrc/gemm_kernel.hpp
    Tile::dma) instead of through registers: same LDS image, same k order, so the same bits (profiles/r05_glds_ab.md is the timing A/B)."""
    from rec_now_amd import _lib
    lib = _lib.load()
    assert lib.recnow_get_gemm_staging() == 0
    C0, Cx0, R, Rx = _product(dev, M, 128, K, 1, 0, 0, 77 + K, scale=0.05)
    _lib.call('recnow_set_gemm_staging', 1)
    try:
        assert lib.recnow_get_gemm_staging() == 1
        C1, Cx1, _, _ = _product(dev, M, 128, K, 1, 0, 0, 77 + K, scale=0.05)
    finally:
        _lib.call('recnow_set_gemm_staging', 0)
    assert np.array_equal(C0, C1) and np.array_equal(Cx0, Cx1)
    assert np.abs(C1 - R).max() <= 1e-5 * np.abs(R).max() and np.abs(Cx1 - Rx).max() <= 1e-5 * np.abs(Rx).max()


_PAIR_SNIPPET = r'''
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from rec_now_amd import _lib
from test_split_precision_gpu import _product
_lib.call('recnow_set_gemm_precision', 1)
dev = torch.device('cuda:0')
h = hashlib.sha256()
for mul in (0, 1):
    C, Cx, _, _ = _product(dev, 2048, 128, 1024, 0, 1, mul, 31 + mul)
    h.update(C.tobytes()); h.update(Cx.tobytes())
print('SHA', h.hexdigest())
'''


def test_paired_a_loads_are_bit_identical_to_the_two_set_schedule(dev):
    """k_gemm_s3's PAIR form (the A loads of k-tiles 2 j and 2 j + 1 requested together: whole 128-byte lines instead of two half-line requests a k-tile
    apart; csrc/gemm_split.hip) changes WHEN operands are requested, not what is summed in which order: same bits as RECNOW_S3_PAIR=0 (read once per
    process: two subprocesses), with and without the second A operand."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for v in ('1', '0'):
        r = subprocess.run([sys.executable, '-c', _PAIR_SNIPPET % (root, os.path.join(root, 'tests'))], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, RECNOW_S3_PAIR=v), cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[v] = [line for line in r.stdout.splitlines() if line.startswith('SHA')][-1]
    assert out['1'] == out['0']


@pytest.mark.parametrize('mul', [0, 1])
def test_lean_split_kernel_with_operands_off_a_line_boundary(dev, split_mode, mul):
    """The PAIR form of k_gemm_s3 needs rows that start on a 128-byte line (it asks for the two halves of a line back to back); an A operand that starts 64 bytes
    into a line -- a column slice of a wider tensor -- takes the two-set schedule instead (csrc/gemm_split.hip `s3_launch`): same bound against fp64."""
    from rec_now_amd import _lib
    M, N, K, off = 1024, 128, 1024, 16
    rng = np.random.default_rng(77 + mul)
    wide = rng.standard_normal((M, K + 32)).astype(np.float32)
    wide2 = rng.uniform(-1, 1, wide.shape).astype(np.float32)
    Bm = rng.standard_normal((N, K)).astype(np.float32)             # [N][K]: the dT2g layout
    Bx = rng.standard_normal((K, 2)).astype(np.float32)
    Wd, W2d, Bd, Bxd = (torch.from_numpy(v).to(dev) for v in (wide, wide2, Bm, Bx))
    C = torch.empty((M, N), device=dev)
    Cx = torch.empty((M, 2), device=dev)
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_trans = Wd.data_ptr() + 4 * off, K + 32, 0
    assert d.A % 128 == 64
    if mul:
        d.A2, d.a_mode = W2d.data_ptr() + 4 * off, 1
    d.B, d.ldb, d.b_trans = Bd.data_ptr(), K, 1
    d.C, d.ldc = C.data_ptr(), N
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = Bxd.data_ptr(), Cx.data_ptr(), 2, 1, 2, 1, 2
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
    torch.cuda.synchronize()
    A64 = wide[:, off:off + K].astype(np.float64) * (wide2[:, off:off + K].astype(np.float64) if mul else 1.0)
    R, Rx = A64 @ Bm.astype(np.float64).T, A64 @ Bx.astype(np.float64)
    assert np.abs(C.cpu().numpy() - R).max() <= 1e-5 * np.abs(R).max()
    assert np.abs(Cx.cpu().numpy() - Rx).max() <= 1e-5 * np.abs(Rx).max()
