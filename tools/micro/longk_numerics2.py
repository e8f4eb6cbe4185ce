"""The GEMM1 product of layer 0 on bench.py's parity inputs (x * 120) in the two arithmetics against fp64: outliers?"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rec_now_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.load()
M, N, K = 16384, 128, 1024
x, _, _ = bench.synth_batch(M, 3, 0)
A = torch.from_numpy(x * np.float32(bench.CHECK_SCALE)).to(dev)
torch.manual_seed(3)
model = bench.Model()
model(A[:256])
Uw = model.cross.origin_to_sub_kernels[0].detach()          # (N_exp, D, S)
Bm = torch.cat([Uw[0], Uw[1]], dim=1).contiguous()         # (D, 128)
BX = model.cross.gate_layers[0].kernel.detach().contiguous()   # (D, 2)
ref = A.double() @ Bm.double()
refx = A.double() @ BX.double()
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
    for name, got, r in (('x U', C[:, :N], ref), ('x K (side)', CX, refx)):
        e = (got.double() - r).abs()
        idx = np.unravel_index(int(e.argmax().item()), tuple(e.shape))
        print('precision %d %-10s max err %.3g (max|ref| %.3g, rel %.3g)  rms err %.3g  above half max: %d  worst at %s: got %.9g ref %.9g' % (prec, name, e.max().item(), r.abs().max().item(),
              e.max().item() / r.abs().max().item(), e.pow(2).mean().sqrt().item(), int((e > 0.5 * e.max()).sum().item()), idx, got[idx].item(), r[idx].item()))
_lib.call('recnow_set_gemm_precision', 0)
