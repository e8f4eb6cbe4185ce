#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): samples/sec, forward+backward, in-batch pairwise loss + 3-layer DCN-v2
(DCNMixLayer, low-rank 64, 2 experts), B = 65536 rows per GPU, 64 fields x 16-dim = 1024 features, on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one forward + backward pass of the hot path over one resident batch (inputs already in HBM):
    x (B,1024) -> DCNMixLayer(dim_sub_space=64, num_layer=3, num_expert=2) -> MultiDenseLayer(1,1) head -> (B,) scores
      -> pairwise_loss(scores, labels, group_id) -> backward to every weight (+ SUM all-reduce of weight grads, N > 1).
Data-parallel (weak scaling): every rank owns whole groups, the loss is combined with one 2-float all-reduce
(rec_now_amd/dp.py).  Prints ONE JSON line on rank 0.

roofline:     the dominant kernel is the exact-fp32 MFMA GEMM `k_gemm<128,128,..>` (the K = 1024 and K = B products of the
              step; the K = 144 products run in the persistent `k_gemm_shortk`, a kernel of its own in rocprof and in
              `all_gemm`); `achieved` = algorithmic flops (2*M*N*K per launch) / HIP-event time of the launches of the
              busiest GEMM kernel during the timed steps (every 5th launch is timed: 18 launches per step, so every
              launch position is sampled equally), measured by the library's own event hook on the launch stream
              (recnow_prof_*).
cpu_baseline: the oracle (dense O(B^2) reference formulation, torch CPU, oracle/dense_ref.py) timed on this host on a
              bounded sample of the same workload (B_s rows with the same 64 rows/group), rank 0, N = 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 65536
N_FIELD, EMB_DIM = 64, 16
D = N_FIELD * EMB_DIM
SUB, LAYERS, EXPERTS = 64, 3, 2
ROWS_PER_GROUP = 64
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PROF_EVERY = 5                        # time every 5th GEMM launch (18 per step: every launch position gets sampled)
GEMM_TAGS = {1: 'k_gemm<128,128,2,2>', 2: 'k_gemm<128,160,4,1>', 3: 'k_gemm<256,64,4,1>', 4: 'k_gemm<256,32,4,1>',
             5: 'k_gemm_shortk'}          # the library's RN_TAG_*: one per GEMM kernel as rocprof names them


def synth_batch(B, seed, rank=0):
    rng = np.random.default_rng(seed + 1000 * rank)
    x = rng.normal(0.0, 0.05, (B, D)).astype(np.float32)
    groups = rng.integers(0, B // ROWS_PER_GROUP, B).astype(np.float32)      # ids are rank-local -> whole groups per rank
    labels = (rng.random(B) < 0.25).astype(np.float32)
    return x, groups, labels


class Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from rec_now_amd.layers.dcn_mix_layer import DCNMixLayer
        from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
        self.cross = DCNMixLayer(dim_sub_space=SUB, num_layer=LAYERS, num_expert=EXPERTS)
        self.head = MultiDenseLayer(1, 1)

    def forward(self, x):
        return self.head(self.cross(x)).reshape(-1)


def cpu_baseline(seconds_budget=20.0):
    """Reference formulation (dense (B,B) masks) on the host CPU, fwd+bwd, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dense_ref as R
    Bs = 8192
    x, groups, labels = synth_batch(Bs, 3)
    g = torch.Generator().manual_seed(3)
    lim = lambda *s: float(np.sqrt(6.0 / (s[-2] + s[-1])))     # noqa: E731
    U = [((torch.rand(EXPERTS, D, SUB, generator=g) * 2 - 1) * lim(D, SUB)).requires_grad_(True) for _ in range(LAYERS)]
    V = [((torch.rand(EXPERTS, SUB, SUB, generator=g) * 2 - 1) * lim(SUB, SUB)).requires_grad_(True) for _ in range(LAYERS)]
    W = [((torch.rand(EXPERTS, SUB, D, generator=g) * 2 - 1) * lim(SUB, D)).requires_grad_(True) for _ in range(LAYERS)]
    b = [torch.zeros(1, EXPERTS, D, requires_grad=True) for _ in range(LAYERS)]
    K = [((torch.rand(D, EXPERTS, generator=g) * 2 - 1) * lim(D, EXPERTS)).requires_grad_(True) for _ in range(LAYERS)]
    hk = ((torch.rand(1, D, 1, generator=g) * 2 - 1) * lim(D, 1)).requires_grad_(True)
    hb = torch.zeros(1, 1, 1, requires_grad=True)
    xt, gt, yt = torch.from_numpy(x), torch.from_numpy(groups), torch.from_numpy(labels)
    params = U + V + W + b + K + [hk, hb]

    def step():
        for p in params:
            p.grad = None
        s = R.multi_dense_layer(R.dcn_mix_layer(xt, U, V, W, b, K), hk, hb).reshape(-1)
        loss = R.pairwise_loss(s, yt, gt)
        loss.backward()
        return float(loss)

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 50:
            break
    return {'value': Bs * n / el, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d steps of fwd+bwd at B=%d (1/8 of the batch, same %d rows/group, same model): dense O(B^2) reference '
                      'formulation restated on torch-CPU fp32 (oracle/dense_ref.py); TF2 itself is not installable here'
                      % (n, Bs, ROWS_PER_GROUP)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='do not record per-launch HIP events in the timed region')
    ap.add_argument('--graph', action='store_true', help='capture the step into a HIP graph and replay it (SURVEY 8f.1; launch-bound small batches)')
    ap.add_argument('--no-input-grad', action='store_true', help='diagnostic: x is data without a gradient (the metric keeps d loss / d x: in a model x is the embedding output)')
    ap.add_argument('--rows', type=int, default=B_PER_GPU, help='rows per GPU (diagnostics; the metric is defined at 65536)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the RCCL process group even at world size 1 (exercises the N>1 code path on one GPU)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)

    from rec_now_amd import _lib, dp
    from rec_now_amd.rec_block.pairwise_loss_from_batch import group_rows, pairwise_loss_fused
    lib = _lib.load()
    dp.FORCE_COLLECTIVES = bool(args.force_dist)

    torch.manual_seed(3)                      # identical replicated weights on every rank
    model = Model()
    rows = args.rows
    x, groups, labels = synth_batch(rows, 3, rank)
    xd, gd, yd = (torch.from_numpy(v).to(dev) for v in (x, groups, labels))
    model(xd[:256])                           # lazy build on the device
    # The step includes the gradient w.r.t. x (in a model x is the concatenated embedding output and needs it); a data-only
    # x lets DCNMixLayer drop every dx product (--no-input-grad, reported as a diagnostic only).
    xd.requires_grad_(not args.no_input_grad)
    params = [p for p in model.parameters()]
    reducer = dp.GradientAllReducer(params)

    side = torch.cuda.Stream(device=dev)

    def step():
        for p in params:
            p.grad = None
        xd.grad = None
        # the grouping of the batch (sort by group id, segments) does not depend on the scores: it runs on a side stream
        # under the forward pass.  Still part of the step: the group ids are an input of every step.
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            seg = group_rows(gd)
        scores = model(xd)
        main.wait_stream(side)
        local_sum, n_pair = pairwise_loss_fused(scores, yd, gd, reduce_mean=False, segments=seg)
        if use_dist:
            # one collective per step: backward on the unnormalised local sum; (loss sum, P) ride in the gradient bucket and
            # the gradients are divided by P_global afterwards (the loss is linear in 1/P) -- no sync between fwd and bwd
            local_sum.backward()
            loss_val, _ = reducer.all_reduce_with_loss(local_sum, n_pair)
        else:
            loss_bw, loss_val, _ = dp.global_pairwise_loss(local_sum, n_pair)
            loss_bw.backward()
        return loss_val

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    graph = None
    if args.graph:
        if use_dist:
            raise SystemExit('--graph is a single-GPU diagnostic')
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_loss = step()
        run_step = lambda: (graph.replay(), graph_loss)[1]     # noqa: E731
        for _ in range(2):
            run_step()
    else:
        run_step = step
    prof = not args.no_prof and graph is None
    if prof:
        _lib.check(lib.recnow_prof_enable(64 * (args.steps + 1)), 'recnow_prof_enable')
        _lib.check(lib.recnow_prof_sample_every(PROF_EVERY), 'recnow_prof_sample_every')
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
    sync()
    elapsed = time.perf_counter() - t0
    roofline = None
    if prof:
        cnt = (ctypes.c_int * 8)()
        ms = (ctypes.c_double * 8)()
        fl = (ctypes.c_double * 8)()
        _lib.check(lib.recnow_prof_collect(cnt, ms, fl), 'recnow_prof_collect')
        lib.recnow_prof_enable(0)
        tag = max(GEMM_TAGS, key=lambda t: ms[t])
        if cnt[tag] > 0:
            achieved = fl[tag] / (ms[tag] * 1e-3) / 1e12
            traffic = None                # HBM bytes per launch from the committed PMC passes (profiles/traffic.json)
            try:
                with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as fh:
                    traffic = json.load(fh)['hbm_bytes_per_launch'].get(GEMM_TAGS[tag])
            except (OSError, ValueError, KeyError):
                pass
            roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic, 'kernel': GEMM_TAGS[tag],
                        'launches': cnt[tag], 'sampled': 'every %dth launch of the timed region' % PROF_EVERY, 'avg_launch_us': ms[tag] * 1e3 / cnt[tag],
                        'algorithmic_flops_per_launch': fl[tag] / cnt[tag],
                        'all_gemm': {GEMM_TAGS[t]: {'launches': cnt[t], 'ms': ms[t],
                                                    'tflops': (fl[t] / (ms[t] * 1e-3) / 1e12) if ms[t] > 0 else None}
                                     for t in GEMM_TAGS if cnt[t] > 0}}
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        out = {
            'metric': 'samples/sec fwd+bwd, in-batch pairwise + DCN-v2, B=65536 at 1/2/4/8 GPUs',
            'value': rows * world * args.steps / elapsed,
            'unit': 'samples/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'configs[2]: dcn_mix_layer (3 cross layers, low-rank 64, 2 experts) + MultiDense(1,1) head + '
                                   'in-batch pairwise (logistic), B=65536 rows per GPU, 64 fields x 16-dim, ~64 rows/group',
                       'global_batch': rows * world, 'input_grad': not args.no_input_grad, 'hip_graph': bool(args.graph), 'parallelism': 'dp%d' % world,
                       'loss': float(loss.item())},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which is flushed at exit, i.e. after a Python print: flush the C
        # streams first so that the JSON line is the LAST line on stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
