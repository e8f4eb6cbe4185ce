"""GPU box: run-to-run determinism of the step at tiny batches (a flaky parity failure at B = 1 was seen once): N steps, every output compared with the
first step's bit for bit.  usage: python tools/micro/stress_small_step.py [B] [D] [L] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ('', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from test_step_gpu import _model  # noqa: E402
from rec_now_amd.step import DCNMixPairwiseStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
dev = torch.device('cuda:0')
x, groups, labels, xd, yd, gd, cross, head = _model(dev, max(B, 256), D, 64, 2, L, 77 + B)
xd, yd = xd[:B].contiguous(), yd[:B].contiguous()
gd = torch.from_numpy((groups[:B] % 7)).to(dev)
# disturb the allocator / leave garbage behind, as a long test session does
junk = [torch.full((1 << 20,), float(i), device=dev) for i in range(8)]
del junk
step = DCNMixPairwiseStep(cross, head, xd, yd, gd)
step.run()
torch.cuda.synchronize()
ref = [step.scores.clone(), step.dx.clone(), step.loss.clone()] + [g.clone() for g in step.grads]
bad = 0
for it in range(N):
    if it % 50 == 0:      # other work between steps (allocations, a big kernel): timing noise
        t = torch.randn(1 << 22, device=dev).sum()
    step.run()
    torch.cuda.synchronize()
    cur = [step.scores, step.dx, step.loss] + list(step.grads)
    for i, (a, b) in enumerate(zip(cur, ref)):
        if not torch.equal(a, b):
            bad += 1
            print('step %d: output %d differs: max |diff| %.3g (ref max %.3g)' % (it, i, float((a - b).abs().max()), float(b.abs().max())), flush=True)
            break
print('B %d D %d L %d: %d of %d steps differ from the first' % (B, D, L, bad, N))
