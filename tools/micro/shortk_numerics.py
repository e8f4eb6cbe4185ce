"""Numerics of the short-K product (K = 144, 130 valid) in the two arithmetics against fp64: max and rms error relative to max|ref|.  GPU only."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rec_now_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.load()
torch.manual_seed(0)
M, N, K, KV = 16384, 1024, 144, 130
for name, scale_rows in (('uniform rows', False), ('rows over 6 decades', True)):
    A = torch.randn(M, K, device=dev)
    A[:, KV:] = 0
    if scale_rows:
        A *= torch.pow(10.0, torch.rand(M, 1, device=dev) * 6 - 3)
    for b_trans in (0, 1):
        Bm = torch.randn(N, K, device=dev) * 0.1 if b_trans else torch.randn(K, N, device=dev) * 0.1
        if b_trans:
            Bm[:, KV:] = 0
        else:
            Bm[KV:, :] = 0
        E = torch.randn(M, N, device=dev)
        ref = (A.double() @ (Bm.double().t() if b_trans else Bm.double())) * E.double()
        for prec in (0, 1):
            _lib.call('recnow_set_gemm_precision', prec)
            C = torch.empty(M, N, device=dev)
            d = _lib.GemmDesc()
            d.A, d.lda, d.a_trans = A.data_ptr(), K, 0
            d.B, d.ldb, d.b_trans = Bm.data_ptr(), (K if b_trans else N), b_trans
            d.C, d.ldc = C.data_ptr(), N
            d.M, d.N, d.K, d.batch = M, N, K, 1
            d.k_valid = KV
            d.emul, d.lde, d.e_mode = E.data_ptr(), N, 1
            ws = _lib.workspace(max(lib.recnow_gemm_workspace_bytes(ctypes.byref(d)), 1 << 20), dev)
            _lib.call('recnow_gemm', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.stream())
            torch.cuda.synchronize()
            err = (C.double() - ref).abs()
            rowrel = (err.max(dim=1).values / ref.abs().max(dim=1).values).max().item()
            print('%-20s b_trans %d precision %d: max err / max|ref| %.3g   rms err / rms ref %.3g   worst row-relative %.3g'
                  % (name, b_trans, prec, err.max().item() / ref.abs().max().item(), (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), rowrel))
_lib.call('recnow_set_gemm_precision', 0)
