"""Times k_mix_chain_fwd (csrc/dcnmix_chain.hip) alone at the c3 shape and checks it against a float64 product."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rec_now_amd import _lib

lib = _lib.load()
B, D, LDT = 65536, 1024, 144
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
T2g = torch.randn(B, LDT, device=dev, generator=g)
Wc2 = torch.randn(LDT, D, device=dev, generator=g) * 0.05
x0 = torch.randn(B, D, device=dev, generator=g)
Wc1 = torch.randn(D, LDT, device=dev, generator=g) * 0.03
gate = torch.randn(D, 2, device=dev, generator=g) * 0.03
out, O = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
T1 = torch.zeros(B, LDT, device=dev)
P = ctypes.c_void_p
fn = lib.recnow_dbg_mix_chain_fwd
fn.restype = ctypes.c_int
fn.argtypes = [P] * 8 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]


def run():
    rc = fn(T2g.data_ptr(), Wc2.data_ptr(), x0.data_ptr(), out.data_ptr(), O.data_ptr(), Wc1.data_ptr(), gate.data_ptr(), T1.data_ptr(),
            B, D, LDT, 2, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print('k_mix_chain_fwd: %.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
if True:
    Tt = T2g.clone()
    Tt[:, 130:] = 0
    Oref = Tt.double() @ Wc2.double()
    outref = Oref * x0.double()
    t1 = torch.tanh(outref @ Wc1[:, :128].double())
    gl = outref @ gate.double()
    print('max err O %.3g out %.3g T1 %.3g gate %.3g' % ((O - Oref).abs().max().item(), (out - outref).abs().max().item(),
                                                          (T1[:, :128] - t1).abs().max().item(), (T1[:, 128:130] - gl).abs().max().item()))
