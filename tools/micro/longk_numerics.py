"""Numerics of the long-K product with a side product (GEMM1 shape, K = 1024) in the two arithmetics against fp64, incl. the SIGN of the error
(a truncating accumulation shows as a bias of err * sign(ref)).  GPU only."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rec_now_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.load()
torch.manual_seed(0)
M, N, K = 16384, 128, 1024
for name, pos in (('zero-mean operands', False), ('positive operands (no cancellation)', True)):
    A = torch.randn(M, K, device=dev) * 0.3
    Bm = torch.randn(K, N, device=dev) * 0.05
    if pos:
        A, Bm = A.abs(), Bm.abs()
    BX = torch.randn(K, 2, device=dev) * 0.05
    ref = A.double() @ Bm.double()
    for prec in (0, 1):
        _lib.call('recnow_set_gemm_precision', prec)
        C = torch.empty(M, 132, device=dev)
        CX = torch.empty(M, 2, device=dev)
        d = _lib.GemmDesc()
        d.A, d.lda, d.a_trans = A.data_ptr(), K, 0
        d.B, d.ldb, d.b_trans = Bm.data_ptr(), N, 0
        d.C, d.ldc = C.data_ptr(), 132
        d.M, d.N, d.K, d.batch = M, N, K, 1
        d.sp_bx, d.sp_cx, d.sp_bx_ks, d.sp_bx_rs, d.sp_cx_ms, d.sp_cx_rs, d.sp_r = BX.data_ptr(), CX.data_ptr(), 2, 1, 2, 1, 2
        ws = _lib.workspace(max(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), 1 << 20), dev)
        _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
        torch.cuda.synchronize()
        e = C[:, :N].double() - ref
        print('%-38s precision %d: max |err| / max|ref| %.3g   rms err / rms ref %.3g   mean(err * sign(ref)) / rms ref %.3g (bias)'
              % (name, prec, e.abs().max().item() / ref.abs().max().item(), (e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                 ((e * ref.sign()).mean() / ref.pow(2).mean().sqrt()).item()))
_lib.call('recnow_set_gemm_precision', 0)
