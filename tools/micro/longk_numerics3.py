"""GEMM1 of every cross layer (x_l U | side product x_l K = gate logits) on bench.py's parity inputs, full batch, in the two arithmetics against the
fp64 product of the SAME fp32 operands: error of the sub-space pre-activations, of the gate logits, and of the softmax gates they give; rows of interest
(argv) are printed one by one.  x_l comes from the fp64 oracle (rounded to fp32), so each layer's product is measured by itself.  GPU only."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle'):
    sys.path.insert(0, p)
import bench
import dense_ref as R
from rec_now_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.load()
rows_of_interest = [int(a) for a in sys.argv[1:]] or [47918, 13342]
M, N, K = 65536, 128, 1024
torch.set_num_threads(2 * bench.cpu_share())
x, _, _ = bench.synth_batch(M, 3, 0)
x = x * np.float32(bench.CHECK_SCALE)
torch.manual_seed(3)
model = bench.Model()
A0 = torch.from_numpy(x).to(dev)
model(A0[:256])
named = {'cross.' + k: v.detach().cpu().double() for k, v in model.cross.named_weights().items()}
named['head.kernel'], named['head.bias'] = model.head.kernel.detach().cpu().double(), model.head.bias.detach().cpu().double()
U, V, W, b, Kg, hk, hb = bench._split(named)
x64 = torch.from_numpy(x).double()
for l in range(3):
    if l == 0:
        xl = x64
    else:
        with torch.no_grad():
            xl = torch.cat([R.dcn_mix_layer(x64[lo:lo + 8192], U[:l], V[:l], W[:l], b[:l], Kg[:l]) for lo in range(0, M, 8192)])
    A = xl.float().to(dev).contiguous()
    Uw = model.cross.origin_to_sub_kernels[l].detach()
    Bm = torch.cat([Uw[0], Uw[1]], dim=1).contiguous()
    BX = model.cross.gate_layers[l].kernel.detach().contiguous()
    ref = (A.double() @ Bm.double())
    refx = (A.double() @ BX.double())
    g_ref = torch.softmax(refx, dim=-1)
    print('layer %d: max|x_l| %.3g  max|x U| %.3g  max|logit| %.3g' % (l, A.abs().max().item(), ref.abs().max().item(), refx.abs().max().item()))
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
        e = (C[:, :N].double() - ref).abs()
        ex = (CX.double() - refx).abs()
        # what matters downstream: tanh of the pre-activation, softmax of the logits
        et = (torch.tanh(C[:, :N].double()) - torch.tanh(ref)).abs()
        eg = (torch.softmax(CX.double(), dim=-1) - g_ref).abs()
        print('  precision %d: x U max err %.3g rms %.3g | tanh(x U) max err %.3g rms %.3g | logits max err %.3g rms %.3g | gates max err %.3g rms %.3g (worst row %d)'
              % (prec, e.max().item(), e.pow(2).mean().sqrt().item(), et.max().item(), et.pow(2).mean().sqrt().item(), ex.max().item(), ex.pow(2).mean().sqrt().item(),
                 eg.max().item(), eg.pow(2).mean().sqrt().item(), int(eg.max(dim=1).values.argmax().item())))
        for r in rows_of_interest:
            print('     row %5d: tanh err max %.3g | logits ref %s err %s | gate ref %.6f err %+.3g'
                  % (r, et[r].max().item(), refx[r].tolist(), (CX[r].double() - refx[r]).tolist(), g_ref[r, 0].item(), (torch.softmax(CX[r].double(), dim=-1)[0] - g_ref[r, 0]).item()))
_lib.call('recnow_set_gemm_precision', 0)
