"""Per-row measurements of the other hot-path kernels at BASELINE config sizes (1x MI355X), fwd+bwd, inputs resident.
Reports the figure each kernel's roofline is priced in: HBM GB/s for FM / DCN-v1 / MoE mix,
rows/s and pairs/s for the ranking losses, TFLOP/s for CIN / MMoE / PLE.   usage: python tools/layer_bench.py [reps] [fm,dcn,pair,list,cin,ple,ipnn,senet,attn,focal,embed]"""
import os
import sys

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')      # see bench.py: ROCm 7.0 graph packet capture + eager launches of the same kernels

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn, n=reps, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n          # ms


def timeit_graph(fn, n=reps):
    """The same step replayed from a captured HIP graph: no Python or launch overhead between kernels, i.e. the GPU time of
    the step.  Returns None when the step cannot be captured (data-dependent host work)."""
    if os.environ.get('RECNOW_LB_NOGRAPH') == '1':      # under rocprofv3: the eager launches are what gets traced
        return None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return timeit(g.replay, n)
    except Exception as e:          # noqa: BLE001
        torch.cuda.synchronize()
        print('   (graph capture not possible: %s)' % repr(e)[:120])
        return None


def fm():
    from rec_now_amd.layers.fm_layer import FMLayer
    B, F, D = 131072, 64, 16                # config 4 global batch
    xs = [torch.randn(B, D, device=dev, requires_grad=True) for _ in range(F)]
    layer = FMLayer()
    gy = torch.randn(B, 1, device=dev)

    def step():
        for x in xs:
            x.grad = None
        layer(xs).backward(gy)
    ms = timeit(step)
    print('FMLayer fwd+bwd   B=%d F=%d D=%d : %.3f ms  %.0f GB/s algorithmic (12*B*F*D bytes)  %.1f M samples/s'
          % (B, F, D, ms, 12.0 * B * F * D / ms / 1e6, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (no host time between the 2 kernels and autograd\'s 64 leaves): %.3f ms  %.0f GB/s'
              % (mg, 12.0 * B * F * D / mg / 1e6))


def dcn():
    from rec_now_amd.layers.dcn_layer import DCNLayer
    B, D, L = 65536, 1024, 3
    x = torch.randn(B, D, device=dev, requires_grad=True)
    layer = DCNLayer(L)
    gy = torch.randn(B, D, device=dev)
    layer(x)

    def step():
        x.grad = None
        layer(x).backward(gy)
    ms = timeit(step)
    print('DCNLayer fwd+bwd  B=%d D=%d L=%d : %.3f ms  %.0f GB/s algorithmic (20*B*D bytes: x,y | x,dy,dx)  %.1f M samples/s'
          % (B, D, L, ms, 20.0 * B * D / ms / 1e6, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.0f GB/s' % (mg, 20.0 * B * D / mg / 1e6))


def zipf_groups(rng, B, a=1.2, cap=2048):
    """SURVEY 8d config 2, skewed variant: group sizes ~ Zipf(a) capped at `cap`, rows shuffled."""
    sizes = []
    while sum(sizes) < B:
        sizes.append(int(min(rng.zipf(a), cap, B - sum(sizes))))
    g = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(g)
    return g.astype(np.float32), len(sizes)


def pairwise(B, G, tag):
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss
    rng = np.random.default_rng(2)
    if G == 0:
        gn, G = zipf_groups(rng, B)
        g = torch.from_numpy(gn).to(dev)
    else:
        g = torch.from_numpy(rng.integers(0, G, B).astype(np.float32)).to(dev)
    y = torch.from_numpy((rng.random(B) < 0.25).astype(np.float32)).to(dev)
    s = torch.randn(B, device=dev, requires_grad=True)
    npair = [0.0]

    def step():
        s.grad = None
        loss, n = pairwise_loss(s, y, g, return_num_pair=True)
        loss.backward()
        npair[0] = n
    ms = timeit(step)
    P = float(npair[0].item())
    print('pairwise_loss %s B=%d groups=%d pairs=%d : %.3f ms  %.1f M rows/s  %.1f M pairs/s' % (tag, B, G, P, ms, B / ms / 1e3, P / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.1f M rows/s' % (mg, B / mg / 1e3))


def listwise():
    from rec_now_amd.rec_block.listwise_loss_from_batch import listwise_loss_from_batch
    B, G = 262144, 4096
    rng = np.random.default_rng(5)
    g = torch.from_numpy(rng.integers(0, G, B).astype(np.float32)).to(dev)
    y = torch.from_numpy((rng.random(B) < 0.25).astype(np.float32)).to(dev)
    s = torch.randn(B, device=dev, requires_grad=True)

    def step():
        s.grad = None
        listwise_loss_from_batch(g, y, s).backward()
    ms = timeit(step)
    print('listwise (fused)  B=%d groups=%d : %.3f ms  %.1f M rows/s' % (B, G, ms, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.1f M rows/s' % (mg, B / mg / 1e3))


def cin():
    from rec_now_amd.layers.cin_layer import CINLayer
    B, F, D, Hs = 16384, 64, 16, [128, 128, 128]          # config 4, one rank's share
    xs = [torch.randn(B, D, device=dev) * 0.1 for _ in range(F)]
    for x in xs:
        x.requires_grad_(True)
    layer = CINLayer(Hs)
    gy = torch.randn(B, D, device=dev)
    layer(xs)

    def step():
        for x in xs:
            x.grad = None
        layer(xs).backward(gy)
    ms = timeit(step, n=max(3, reps // 3))
    ext = [F] + Hs
    fwd = 2.0 * D * F * sum(ext[k - 1] * ext[k] for k in range(1, len(ext))) * B
    import os
    n = 4 if os.environ.get('RECNOW_CIN_FUSED') == '0' else 3       # round 4: dX_{k-1} and dx0 come out of ONE forward-sized product (csrc/cin_bwd.hip)
    print('CINLayer fwd+bwd  B=%d F=%d D=%d H=%s : %.2f ms  %.1f TFLOP/s executed (%dx fwd flops: 1 fwd + %d bwd products), %.1f TFLOP/s of the 3x-fwd algorithmic flops  %.1f k samples/s'
          % (B, F, D, Hs, ms, n * fwd / ms / 1e9, n, n - 1, 3 * fwd / ms / 1e9, B / ms))


def ple():
    from rec_now_amd.layers.ple_layer import PLELayer
    B, Din = 32768, 4096                                 # config 5, one rank's share (SURVEY-chosen dims)
    x = torch.randn(B, Din, device=dev) * 0.05
    layer = PLELayer(3, [[512, 256], [256, 128]], 2, 1, activation='relu')
    layer(x)

    def step():
        for p in layer.parameters():
            p.grad = None
        outs = layer(x)
        sum(o.sum() for o in outs).backward()
    ms = timeit(step, n=max(3, reps // 3))
    fl = 0.0
    for p in layer.parameters():
        if p.dim() == 3:
            fl += 2.0 * B * p.numel()
        elif p.dim() == 2:
            fl += 2.0 * B * p.numel()
    print('PLELayer fwd+bwd  B=%d Din=%d 3 tasks : %.2f ms  ~%.1f TFLOP/s (3x fwd GEMM flops)  %.1f k samples/s' % (B, Din, ms, 3 * fl / ms / 1e9, B / ms))


def ipnn():
    from rec_now_amd.layers.inner_pnn_layer import InnerPNNLayer
    B, F, D = 131072, 64, 16
    xs = [torch.randn(B, D, device=dev, requires_grad=True) for _ in range(F)]
    P = F * (F - 1) // 2
    gy = torch.randn(B, P, device=dev)
    layer = InnerPNNLayer()

    def step():
        for x in xs:
            x.grad = None
        layer(xs).backward(gy)
    ms = timeit(step)
    print('InnerPNN fwd+bwd  B=%d F=%d D=%d P=%d : %.3f ms  %.0f GB/s algorithmic (8*B*P + 12*B*F*D bytes)  %.1f M samples/s'
          % (B, F, D, P, ms, (8.0 * B * P + 12.0 * B * F * D) / ms / 1e6, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.0f GB/s' % (mg, (8.0 * B * P + 12.0 * B * F * D) / mg / 1e6))


def senet():
    from rec_now_amd.layers.senet_layer import SENETLayer
    B, F, D = 131072, 64, 16
    xs = [torch.randn(B, D, device=dev, requires_grad=True) for _ in range(F)]
    gy = torch.randn(B, F * D, device=dev)
    layer = SENETLayer(0.25)
    layer(xs)

    def step():
        for x in xs:
            x.grad = None
        layer(xs).backward(gy)
    ms = timeit(step)
    print('SENETLayer fwd+bwd B=%d F=%d D=%d : %.3f ms  %.0f GB/s algorithmic (24*B*F*D bytes: x x2 fwd, x dout x2 + dx bwd)  %.1f M samples/s'
          % (B, F, D, ms, 24.0 * B * F * D / ms / 1e6, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.0f GB/s' % (mg, 24.0 * B * F * D / mg / 1e6))


def attn():
    from rec_now_amd.rec_block.attention import attention_by_dot_product
    B, L, D = 131072, 50, 16
    u = torch.randn(B, L, D, device=dev, requires_grad=True)
    d = torch.randn(B, D, device=dev, requires_grad=True)
    gm = torch.randn(B, D, device=dev)

    def step():
        u.grad = None
        d.grad = None
        mat, s = attention_by_dot_product(u, d, filter_neg=True)
        (mat * gm).sum().add(s.sum()).backward()
    ms = timeit(step)
    print('attention_by_dot_product fwd+bwd B=%d L=%d D=%d : %.3f ms  %.0f GB/s algorithmic (12*B*L*D bytes)  %.1f M samples/s'
          % (B, L, D, ms, 12.0 * B * L * D / ms / 1e6, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.0f GB/s' % (mg, 12.0 * B * L * D / mg / 1e6))


def focal():
    from rec_now_amd.rec_block.focal_loss import focal_crossentropy_loss
    B = 1 << 22
    x = torch.randn(B, device=dev, requires_grad=True)
    z = (torch.rand(B, device=dev) < 0.25).float()

    def step():
        x.grad = None
        focal_crossentropy_loss(z, x).backward()
    ms = timeit(step)
    print('focal_crossentropy_loss fwd+bwd B=%d : %.3f ms  %.0f GB/s algorithmic (20*B bytes)  %.0f M samples/s' % (B, ms, 20.0 * B / ms / 1e6, B / ms / 1e3))
    mg = timeit_graph(step)
    if mg:
        print('   replayed from a HIP graph (GPU time of the step): %.3f ms  %.0f GB/s' % (mg, 20.0 * B / mg / 1e6))


def embed():
    from rec_now_amd.rec_block.embedding_util import EmbeddingTable, embedding_using_sparse_batch_segment_ids
    B, C, T, D, V = 65536, 100, 64, 16, 1 << 20          # c3: 64 pooled fields x 16-dim from 100 id columns per row
    rng = np.random.default_rng(7)
    slots = torch.from_numpy(rng.integers(0, 80, (B, C)).astype(np.int32)).to(dev)
    ids = torch.from_numpy((rng.zipf(1.3, (B, C)) % V).astype(np.int64)).to(dev)
    table = EmbeddingTable(torch.randn(V, D, device=dev) * 0.05)
    gy = torch.randn(B, T, D, device=dev)
    targets = list(range(T))

    def fwd():
        return embedding_using_sparse_batch_segment_ids(table, slots, targets, ids)

    def step():
        table.weight.grad = None
        fwd().backward(gy)
    ms_f = timeit(fwd)
    ms = timeit(step)
    pooled = float((slots < T).sum().item())
    print('embedding pooled lookup B=%d C=%d T=%d D=%d V=%d : fwd %.3f ms (%.0f GB/s: 4*D bytes gathered per pooled id + 16 B/id + 4*B*T*D out), '
          'fwd+bwd %.3f ms (sort by id + per-id reduction + dense scatter)  %.1f M ids/s'
          % (B, C, T, D, V, ms_f, (pooled * 4 * D + 16.0 * B * C + 4.0 * B * T * D) / ms_f / 1e6, ms, B * C / ms / 1e3))


if __name__ == '__main__':
    which = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fm', 'dcn', 'pair', 'list', 'cin', 'ple', 'ipnn', 'senet', 'attn', 'focal', 'embed']
    if 'fm' in which:
        fm()
    if 'dcn' in which:
        dcn()
    if 'pair' in which:
        pairwise(8192, 128, 'config2')
        pairwise(8192, 0, 'config2-skewed (Zipf 1.2 group sizes, cap 2048)')
        pairwise(65536, 1024, 'config3')
        pairwise(65536, 0, 'config3-skewed (Zipf 1.2 group sizes, cap 2048)')
    if 'list' in which:
        listwise()
    if 'cin' in which:
        cin()
    if 'ple' in which:
        ple()
    for name, fn in (('ipnn', ipnn), ('senet', senet), ('attn', attn), ('focal', focal), ('embed', embed)):
        if name in which:
            fn()
