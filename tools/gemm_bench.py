"""Microbenchmark of recnow_gemm on the shapes the DCN-v2 step uses (+ a square reference).  GPU only.
usage: python tools/gemm_bench.py [reps] [shape_index]"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rec_now_amd import _lib

dev = torch.device('cuda:0')
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, D, KC, LDT = 65536, 1024, 130, 132

SHAPES = [
    # name, M, N, K, a_trans, b_trans, lda, ldb, ldc, a_mode, emul, c_trans
    ('GEMM1  x[U|K]        (B,D)x(D,130)', B, KC, D, 0, 0, D, LDT, LDT, 0, 0, 0),
    ('GEMM3  x*(T2g[W;b])  (B,130)x(130,D)', B, D, KC, 0, 0, LDT, D, D, 0, 1, 0),
    ('dT2g   (x*g)Wc2^T    (B,D)x(D,130)', B, KC, D, 0, 1, D, D, LDT, 1, 0, 0),
    ('dxl    dT1 Wc1^T     (B,130)x(130,D)', B, D, KC, 0, 1, LDT, LDT, D, 0, 0, 0),
    ('dWc1   xl^T dT1      (D,B)x(B,130)', D, KC, B, 1, 0, D, LDT, LDT, 0, 0, 0),
    ('dWc2^T (x*g)^T T2g   (D,B)x(B,130)', D, KC, B, 1, 0, D, LDT, D, 1, 0, 1),
    # tile-aligned (padded) versions of the same products: N 130 -> 160 columns, K 130 -> 144, leading dim 160
    ('pad GEMM1  (B,D)x(D,160)', B, 160, D, 0, 0, D, 160, 160, 0, 0, 0),
    ('pad GEMM3  (B,144)x(144,D)', B, D, 144, 0, 0, 160, D, D, 0, 1, 0),
    ('pad dT2g   (B,D)x(D,160) A*A2', B, 160, D, 0, 1, D, D, 160, 1, 0, 0),
    ('pad dxl    (B,144)x(144,D)', B, D, 144, 0, 1, 160, 160, D, 0, 0, 0),
    ('pad dWc1   (D,B)x(B,160)', D, 160, B, 1, 0, D, 160, 160, 0, 0, 0),
    ('pad dWc2^T (D,B)x(B,160) A*A2', D, 160, B, 1, 0, D, 160, D, 1, 0, 1),
    # exact-128 versions (no side product here: pure MFMA part)
    ('x128 GEMM1  (B,D)x(D,128)', B, 128, D, 0, 0, D, 128, 132, 0, 0, 0),
    ('x128 GEMM3  (B,128)x(128,D)', B, D, 128, 0, 0, 132, D, D, 0, 1, 0),
    ('x128 dT2g   (B,D)x(D,128) A*A2', B, 128, D, 0, 1, D, D, 132, 1, 0, 0),
    ('x128 dxl    (B,128)x(128,D)', B, D, 128, 0, 1, 132, 128, D, 0, 0, 0),
    ('x128 dU     (D,B)x(B,128)', D, 128, B, 1, 0, D, 132, 128, 0, 0, 0),
    ('x128 dW^T   (D,B)x(B,128) A*A2', D, 128, B, 1, 0, D, 132, D, 1, 0, 1),
    ('x128+SP GEMM1', B, 128, D, 0, 0, D, 128, 132, 0, 0, 0, 1),
    ('x128+EU GEMM3', B, D, 128, 0, 0, 132, D, D, 0, 1, 0, 2),
    ('x128+SP dT2g', B, 128, D, 0, 1, D, D, 132, 1, 0, 0, 1),
    ('x128+EU dxl', B, D, 128, 0, 1, 132, 128, D, 0, 0, 0, 2),
    ('x128+SP dU', D, 128, B, 1, 0, D, 132, 128, 0, 0, 0, 1),
    ('x128+SP dW^T', D, 128, B, 1, 0, D, 132, D, 1, 0, 1, 1),
    ('pad160 GEMM3 (B,160)x(160,D) BK32', B, D, 160, 0, 0, 160, D, D, 0, 1, 0),
    ('pad160 dxl   (B,160)x(160,D) BK32', B, D, 160, 0, 1, 160, 160, D, 0, 0, 0),
    ('square 4096^3 NN', 4096, 4096, 4096, 0, 0, 4096, 4096, 4096, 0, 0, 0),
    ('square 4096^3 NT', 4096, 4096, 4096, 0, 1, 4096, 4096, 4096, 0, 0, 0),
]


def run(name, M, N, K, ta, tb, lda, ldb, ldc, a_mode, emul, c_trans, xf=0):
    rows_a = K if ta else M
    rows_b = N if tb else K
    A = torch.randn(rows_a, lda, device=dev) * 0.1
    A2 = torch.randn(rows_a, lda, device=dev) if a_mode else None
    Bm = torch.randn(rows_b, ldb, device=dev) * 0.1
    C = torch.empty((N if c_trans else M), ldc, device=dev)
    E = torch.randn(M, N, device=dev) if emul else None
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_trans = A.data_ptr(), lda, ta
    if a_mode:
        d.A2, d.a_mode = A2.data_ptr(), 1
    d.B, d.ldb, d.b_trans = Bm.data_ptr(), ldb, tb
    d.C, d.ldc, d.c_trans = C.data_ptr(), ldc, c_trans
    d.M, d.N, d.K, d.batch = M, N, K, 1
    if emul:
        d.emul, d.lde, d.e_mode = E.data_ptr(), N, 1
    if xf == 1:
        BX = torch.randn(K, 2, device=dev); CX = torch.empty(M, 2, device=dev)
        d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = BX.data_ptr(), CX.data_ptr(), 2, 1, 2, 1, 2
    if xf == 2:
        PP = torch.randn(M, 2, device=dev); QQ = torch.randn(2, N, device=dev)
        d.eu_p, d.eu_q, d.eu_pms, d.eu_qrs, d.eu_qns, d.eu_r = PP.data_ptr(), QQ.data_ptr(), 2, N, 1, 2
    ws = _lib.workspace(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), dev)
    st = _lib.stream()
    for _ in range(2):
        _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print('%-40s %9.1f us  %6.1f TFLOP/s' % (name, us, 2.0 * M * N * K / us / 1e6), flush=True)


def run_midf(name='MIDF   GEMM1^T + sub-space forward in the epilogue (k_gemm<..,25>)'):
    """The fused GEMM1 of the c3 step (dcnmix.hip: `d.mid_V`): T1^T = [U | K]^T x^T computed transposed, H1 = tanh, C = H1 V_e off the accumulators,
    T1 / T2 / T2g written by the epilogue.  Shape index len(SHAPES) (for tools/gemm_trace.py too)."""
    S, N, LD = 64, 2, 144
    x = torch.randn(B, D, device=dev) * 0.05
    Wc1 = torch.randn(D, LD, device=dev) * 0.03          # packed [U_0 | U_1 | K | 0]
    gate = torch.randn(D, N, device=dev) * 0.03
    V = torch.randn(N, S, S, device=dev) * 0.1
    T1, T2, T2g = (torch.empty(B, LD, device=dev) for _ in range(3))
    d = _lib.GemmDesc()
    d.A, d.lda, d.a_trans = Wc1.data_ptr(), LD, 1
    d.B, d.ldb, d.b_trans = x.data_ptr(), D, 1
    d.C, d.ldc = T1.data_ptr(), LD
    d.M, d.N, d.K, d.batch = N * S, B, D, 1
    d.act = 2                                             # RECNOW_ACT_TANH
    d.sp_bx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = gate.data_ptr(), N, 1, T1.data_ptr() + 4 * N * S, LD, 1, N
    d.mid_V, d.mid_T1, d.mid_T2, d.mid_T2g, d.mid_ld, d.mid_act_outer = V.data_ptr(), T1.data_ptr(), T2.data_ptr(), T2g.data_ptr(), LD, 2
    ws = _lib.workspace(max(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), 256), dev)
    st = _lib.stream()
    for _ in range(2):
        _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print('%-40s %9.1f us  %6.1f TFLOP/s of the product alone' % (name, us, 2.0 * B * (N * S + N) * D / us / 1e6), flush=True)


only = int(sys.argv[2]) if len(sys.argv) > 2 else None
if __name__ == '__main__':
    for i, s in enumerate(SHAPES):
        if only is None or i == only:
            run(*s)
    if only is None or only == len(SHAPES):
        run_midf()
