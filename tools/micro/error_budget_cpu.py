"""Which fp32 rounding of the c3 step's forward chain uses the parity budget?  CPU only (torch fp64 + the C pair oracle).

The whole step (bench.py's parity inputs: x * 120, seed 3) is evaluated in fp64 with ONE quantity at a time replaced by its fp32-accumulated value
(torch CPU sgemm of the fp32-rounded operands: not the GPU's summation order, but the same error scale), and d loss / d x is compared with the all-fp64
result: the contribution of that quantity's rounding to the max-norm error the gate measures.

    python tools/micro/error_budget_cpu.py [rows]
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle'):
    sys.path.insert(0, p)
import pairs_oracle as PO

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, S, N, L = 1024, 64, 2, 3
rng = np.random.default_rng(3)
x = (rng.normal(0.0, 0.05, (B, D)).astype(np.float32) * np.float32(120.0))
groups = rng.integers(0, max(B // 64, 1), B).astype(np.float32)
labels = (rng.random(B) < 0.25).astype(np.float32)
g = torch.Generator().manual_seed(3)
lim = lambda a, b: float(np.sqrt(6.0 / (a + b)))      # noqa: E731
U = [((torch.rand(N, D, S, generator=g) * 2 - 1) * lim(D, S)).double() for _ in range(L)]
V = [((torch.rand(N, S, S, generator=g) * 2 - 1) * lim(S, S)).double() for _ in range(L)]
W = [((torch.rand(N, S, D, generator=g) * 2 - 1) * lim(S, D)).double() for _ in range(L)]
K = [((torch.rand(D, N, generator=g) * 2 - 1) * lim(D, N)).double() for _ in range(L)]
hk = ((torch.rand(D, generator=g) * 2 - 1) * lim(D, 1)).double()


def f32mm(a, b):
    """a @ b with fp32 accumulation of the fp32-rounded operands; the gradient flows as through the fp64 product (a straight-through perturbation)."""
    exact = a @ b
    with torch.no_grad():
        pert = (a.float() @ b.float()).double() - exact
    return exact + pert


def forward(xc, mode):
    cur = xc
    for l in range(L):
        lo = mode.get('logits', ())
        xu = mode.get('xu', ())
        sub = torch.stack([(f32mm(cur, U[l][n]) if l in xu else cur @ U[l][n]) for n in range(N)], 1)       # (b, N, S)
        sub = torch.tanh(sub)
        sub = torch.tanh(torch.einsum('bns,nst->bnt', sub, V[l]))
        org = torch.einsum('bns,nsd->bnd', sub, W[l])
        org = xc.unsqueeze(1) * org
        logits = f32mm(cur, K[l]) if l in lo else cur @ K[l]
        if l in mode.get('logits_err', {}):
            with torch.no_grad():
                noise = mode['logits_err'][l] * torch.randn(logits.shape, dtype=torch.float64, generator=g)
            logits = logits + noise
        gates = torch.softmax(logits, -1)
        cur = torch.einsum('bnd,bn->bd', org, gates)
    return cur @ hk


def step(mode, chunk=4096):
    sc = np.empty(B)
    with torch.no_grad():
        for lo in range(0, B, chunk):
            torch.manual_seed(lo)
            g.manual_seed(1000 + lo)
            sc[lo:lo + chunk] = forward(torch.from_numpy(x[lo:lo + chunk]).double(), mode).numpy()
    _, ds, _ = PO.pairwise_bpr(groups, labels, sc, grouped=True)
    ds = torch.from_numpy(np.asarray(ds, np.float64))
    dx = np.empty((B, D))
    for lo in range(0, B, chunk):
        g.manual_seed(1000 + lo)
        xc = torch.from_numpy(x[lo:lo + chunk]).double().requires_grad_(True)
        forward(xc, mode).backward(ds[lo:lo + chunk])
        dx[lo:lo + chunk] = xc.grad.numpy()
    return sc, dx


torch.set_num_threads(os.cpu_count() or 8)
sc0, dx0 = step({})
m = np.abs(dx0).max()
print('rows %d: max|dx| %.3g  max|score| %.3g' % (B, m, np.abs(sc0).max()))
for name, mode in (('layer-0 gate logits fp32', {'logits': (0,)}), ('layer-1,2 gate logits fp32', {'logits': (1, 2)}), ('layer-0 x U fp32', {'xu': (0,)}),
                   ('layer-1,2 x U fp32', {'xu': (1, 2)}), ('all GEMM1 products fp32', {'logits': (0, 1, 2), 'xu': (0, 1, 2)}),
                   ('layer-0 logits + N(0, 3.5e-6) [the GPU kernels, measured]', {'logits_err': {0: 3.5e-6}}),
                   ('layer-0 logits + N(0, 5e-7) [fp64 running sum]', {'logits_err': {0: 5e-7}})):
    sc, dx = step(mode)
    e = np.abs(dx - dx0)
    print('%-62s scores max err / max %.3g | dx max err / max %.3g  rms err / rms %.3g' % (name, np.abs(sc - sc0).max() / np.abs(sc0).max(), e.max() / m,
          np.sqrt((e ** 2).mean()) / np.sqrt((dx0 ** 2).mean())), flush=True)
