import torch, time
n = 131072*2016
a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda')
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps
ms = t(lambda: a.fill_(1.0)); print('fill 1.06GB', ms, 'ms', n*4/ms/1e6, 'GB/s')
ms = t(lambda: a.copy_(b)); print('copy', ms, 'ms', 2*n*4/ms/1e6, 'GB/s')
ms = t(lambda: torch.sum(a)); print('sum', ms, 'ms', n*4/ms/1e6, 'GB/s')
