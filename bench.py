#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): samples/sec, forward+backward, in-batch pairwise loss + 3-layer DCN-v2
(DCNMixLayer, low-rank 64, 2 experts), global batch B = 65536, 64 fields x 16-dim = 1024 features, on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one forward + backward pass of the hot path over one resident batch (inputs already in HBM):
    x (B,1024) -> DCNMixLayer(dim_sub_space=64, num_layer=3, num_expert=2) -> MultiDenseLayer(1,1) head -> (B,) scores
      -> pairwise_loss(scores, labels, group_id) -> backward to every weight (+ SUM all-reduce of weight grads, N > 1).
Data-parallel: STRONG scaling by default -- the metric's B = 65536 is the global batch, every rank owns 65536 / N rows of whole
groups (`--scaling weak`: 65536 rows per rank); the loss statistics ride in the first gradient bucket (rec_now_amd/dp.py), no
data-path collective.  `--rows R --force-dist` runs the per-rank shard of an N-GPU row on one GPU.  Prints ONE JSON line on rank 0.

roofline:     the dominant kernel is the exact-fp32 MFMA GEMM `k_gemm<128,128,..>` (the K = 1024 and K = B products of the
              step); `achieved` = algorithmic flops (2*M*N*K per launch) / HIP-event time of its launches during the timed
              steps (every 5th launch of the hooked kernels is timed, so every launch position is sampled equally), measured by
              the library's own event hook on the launch stream (recnow_prof_*).  `hbm_bound_kernels` holds the other side: the
              K = 144 products (`k_gemm_shortk`, whose fused epilogues stream three to four (B,D) tensors) and the sub-space
              kernels `k_mix_mid_fwd/bwd` as achieved GB/s of their ALGORITHMIC bytes against the 8 TB/s HBM3E spec.
              `traffic` (PMC bytes per launch) is read from profiles/traffic.json only while that file's hash of the kernel
              sources matches this build; otherwise it is null and `traffic_stale` says so.
parity:       after the timed region ONE more step runs on inputs scaled so that the scores are of O(1) (loss != ln 2; the
              timed inputs give scores ~1e-8, where a library returning zeros would print the same loss) and is held to 1e-5
              (north_star) against the fp64 oracle: (a) ~256 rows of whole groups, scores and d loss / d x; (b) unless
              --no-cpu-baseline: the WHOLE batch -- loss, pair count, scores, d loss / d x and every one of the 17 weight gradients
              against oracle/dense_ref.py evaluated chunk-wise in fp64.  The fp32 CPU port's figures are reported beside.
cpu_baseline: rank 0, N = 1 only.  `value`: ONE step of the same workload at the full batch (B = 65536) by the CPU port:
              oracle/dense_ref.py layers in row chunks (torch CPU fp32, two passes: scores, then forward + backward per chunk
              with the pair gradient) + the segment-based C pair loss (oracle/pairs_oracle.c, OpenMP); the reference's own dense
              (B,B) formulation cannot run at this B (>= 100 GB of temporaries), so its figure at B = 8192 is kept beside it in
              `dense_b8192`.
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time

# HIP graphs on ROCm 7.0: the runtime's graph AQL-packet capture (pre-built dispatch packets replayed straight into the queue) faults --
# "Memory access fault ... write access to a read-only page" -- when a step that was captured as SEVERAL graphs (one per backward piece,
# the N > 1 path) is replayed after the same kernels were launched eagerly in between (reproduced in tools/graph_dist_probe.py: clean with
# the switch below, faults without it; single-graph steps were never affected).  The runtime reads the switch when it is loaded, i.e. at
# `import torch`: INTEGRATION.md tells trainers that replay rec_now_amd.step graphs to export it as well.
import os as _os
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GLOBAL_BATCH = 65536                  # BASELINE.json metric: B = 65536 at 1/2/4/8 GPUs (strong scaling: 65536 / N rows per rank)
B_PER_GPU = 65536                     # --scaling weak: rows per rank
N_FIELD, EMB_DIM = 64, 16
D = N_FIELD * EMB_DIM
SUB, LAYERS, EXPERTS = 64, 3, 2
ROWS_PER_GROUP = 64
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 16 * 157.3     # same guide: the fp32 MFMA rate is 1/16 of the dense bf16 rate (~2.5 PFLOP/s)
PEAK_HBM_GBS = 8000.0                 # same guide, "HBM3E peak BW" (spec; 6.29 TB/s is the measured copy ceiling)
REWARM_STEPS, REWARM_SECONDS = 3, 0.06   # untimed steps between the host-side preparations of the timed region (gc, hook events) and its start: at least 3, and 60 ms of them
N_TAGS = 32                           # host arrays of recnow_prof_collect: at least recnow_prof_tag_count() entries (checked in main)
PROF_EVERY = 5                        # time every 5th hooked launch (24 per step: every launch position gets sampled)
GEMM_TAGS = {1: 'k_gemm<128,128,2,2>', 2: 'k_gemm<128,160,4,1>', 3: 'k_gemm<256,64,4,1>', 4: 'k_gemm<256,32,4,1>',
             5: 'k_gemm_shortk', 8: 'k_gemm_split',         # the library's RN_TAG_*: one per GEMM kernel as rocprof names them
             9: 'k_gemm<128,128,2,2,..,25> (GEMM1 + sub-space forward in its epilogue)',
             10: 'k_mix_tile_fwd (row-block persistent forward of all cross layers, shard sizes)',
             11: 'k_mix_tile_bwd (row-block persistent backward chain of all cross layers, shard sizes)',
             12: 'k_gemm<64,128,1,4> (small-M dispatch of the long-K products, shard sizes)'}
HBM_TAGS = {5: 'k_gemm_shortk', 6: 'k_mix_mid_fwd', 7: 'k_mix_mid_bwd'}
# tags that only the every-launch mode records (csrc/prof.hpp): with them the hooked intervals cover the whole step
PHASE_TAGS = {6: 'k_mix_mid_fwd', 7: 'k_mix_mid_bwd', 13: 'grouping of the batch (keys, radix sort, segments)',
              14: 'loss stage (pair walks, finalize, d loss / d scores)', 15: 'weight packs + layer-end reductions + head post-processing'}
CHECK_SCALE = 120.0                   # the parity step's inputs: x * 120 (std 6) -> scores of O(0.3), loss != ln 2
# element-relative gates of the full-batch comparison: (floor, tolerance) -- every entry of at least `floor` x its tensor's largest magnitude within
# `tolerance` of ITSELF.  An entry at the floor may carry the whole norm-relative 1e-5 of the largest entry, i.e. 1e-5 / floor of itself (fp32 sums of
# 1024 / 65 536 terms cancel: the reference's own fp32 arithmetic has the same property), so the bounds are PARITY_TOL / floor; what is measured sits
# 5-10x inside them (first run: 1.8e-3 at the 1e-3 floor) and is reported per tensor.
ELEM_GATES = ((1e-3, 1e-2), (1e-2, 1e-3))
SPLIT_FROM_ROWS = 32768               # --gemm-precision auto: split-precision products from this many rows per GPU (measured: 1.68 vs 1.82 ms at 32 768 rows, 1.10 vs 1.01 at 16 384)
PARITY_TOL = 1e-5                     # north_star: 1e-5 relative, GPU fp32 against the fp64 oracle (row subset + the full batch)


def cpu_share():
    """CPUs this process may really use (the cgroup quota: a GPU box shows 256 logical CPUs and grants 16).  The host legs of this script -- the CPU baseline
    and the fp64 oracle of the gate -- run on twice that many threads (measured best on such a box: 56 s with 32 threads against 102 s with torch's
    default of 128, which mostly waits for its quota)."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            return max(1, -(-int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def kernel_source_hash():
    """sha256 over the HIP sources of the library: profiles/traffic.json is only valid for the build it was measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'rec_now_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'rec_now_amd', 'csrc', '*.hpp'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def synth_batch(B, seed, rank=0):
    rng = np.random.default_rng(seed + 1000 * rank)
    x = rng.normal(0.0, 0.05, (B, D)).astype(np.float32)
    # every rank draws its own B // 64 group ids and shifts them by rank * (B // 64): whole groups per rank, ids distinct across
    # ranks, so the ranks' shards concatenate to ONE global batch whose loss the N-rank step reproduces (rec_now_amd/dp.py)
    n_groups = max(B // ROWS_PER_GROUP, 1)
    groups = (rng.integers(0, n_groups, B) + rank * n_groups).astype(np.float32)
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


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dense_ref as R
    import pairs_oracle as PO
    return R, PO


def _split(named):
    """named weights of Model -> the oracle's argument lists."""
    pick = lambda stem: [named['cross.%s_of_layer%d' % (stem, l)] for l in range(LAYERS)]      # noqa: E731
    return (pick('origin_to_sub_kernels'), pick('sub_to_sub_kernels'), pick('sub_to_origin_kernels'), pick('bias'),
            [named['cross.gate_of_layer%d/kernel' % l] for l in range(LAYERS)], named['head.kernel'], named['head.bias'])


def _cpu_model(named, dtype):
    """The oracle's layers over the named weights: (forward closure rows -> scores, {name: leaf tensor})."""
    R, _ = _oracle()
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).requires_grad_(True) for k, v in named.items()}
    U, V, W, b, K, hk, hb = _split(w)
    return (lambda xc: R.multi_dense_layer(R.dcn_mix_layer(xc, U, V, W, b, K), hk, hb).reshape(-1)), w


def cpu_scores(fwd, x, dtype, chunk=8192):
    """Scores of the rows of x by the oracle's layers, in row chunks (no gradient)."""
    xt = torch.from_numpy(x)
    scores = np.empty(x.shape[0], np.float32 if dtype == torch.float32 else np.float64)
    with torch.no_grad():
        for lo in range(0, x.shape[0], chunk):
            scores[lo:lo + chunk] = fwd(xt[lo:lo + chunk].to(dtype)).numpy()
    return scores


def cpu_backward(fwd, x, ds, dtype, chunk=8192):
    """Backward of the oracle's layers over the rows of x given d loss / d score: returns d loss / d x; the weight gradients
    accumulate in the leaf tensors of `_cpu_model` (summed over the chunks: the layers are row-separable)."""
    xt = torch.from_numpy(x)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    ds_t = torch.from_numpy(np.asarray(ds).astype(npdt))
    dx = np.empty(x.shape, npdt)
    for lo in range(0, x.shape[0], chunk):
        xc = xt[lo:lo + chunk].to(dtype).clone().requires_grad_(True)
        fwd(xc).backward(ds_t[lo:lo + chunk])
        dx[lo:lo + chunk] = xc.grad.numpy()
    return dx


def cpu_step_full(x, groups, labels, named, chunk=8192, dtype=torch.float32, warm_rows=0):
    """ONE fwd+bwd step of the workload at the full batch on the host: oracle layers in row chunks (torch CPU, `dtype`) + the
    segment-based C pair loss.  dtype float32: the CPU baseline that is timed (after an untimed warm-up over `warm_rows` rows, so the
    figure is not a cold first call); float64: the oracle every output and gradient of the GPU step is held to.
    Returns (loss, dx, {name: grad}, scores, n_pair, seconds)."""
    _, PO = _oracle()
    fwd, w = _cpu_model(named, dtype)
    if warm_rows:
        xc = torch.from_numpy(x[:warm_rows]).to(dtype).requires_grad_(True)
        fwd(xc).sum().backward()
        for v in w.values():
            v.grad = None
    t0 = time.perf_counter()
    scores = cpu_scores(fwd, x, dtype, chunk)
    loss, ds, n_pair = PO.pairwise_bpr(groups, labels, scores, grouped=True)
    dx = cpu_backward(fwd, x, ds, dtype, chunk)
    sec = time.perf_counter() - t0
    return loss, dx, {k: v.grad.numpy() for k, v in w.items()}, scores, n_pair, sec


def cpu_dense_b8192(seconds_budget=8.0):
    """The reference's own dense (B,B) formulation (oracle/dense_ref.py, torch CPU fp32) at the largest B it is practical at."""
    R, _ = _oracle()
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
        R.pairwise_loss(s, yt, gt).backward()

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 50:
            break
    return {'value': Bs * n / el, 'unit': 'samples/s', 'steps': n, 'rows': Bs,
            'what': 'dense O(B^2) reference formulation restated on torch-CPU fp32 (oracle/dense_ref.py), 1/8 of the batch, same 64 rows/group'}


def rel_err(a, b, scale=None):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    s = max(float(np.abs(b).max()) if scale is None else float(scale), 1e-30)
    return float(np.abs(a - b).max() / s)


def rel_err_elem(a, b, floor=1e-3):
    """Element-relative error: max over the entries with |ref| >= floor * max|ref| of |a - ref| / |ref| (the norm-relative `rel_err` lets
    an entry a thousand times below the largest one be wrong in its leading digits; this one does not)."""
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    m = float(np.abs(b).max()) if b.size else 0.0
    if m <= 0.0:
        return 0.0
    keep = np.abs(b) >= floor * m
    return float((np.abs(a[keep] - b[keep]) / np.abs(b[keep])).max())


def subset_parity(x, groups, labels, named, scores_gpu, dx_gpu, n_pair_gpu, n_groups=4, p_total=None):
    """~256 rows of whole groups of the full-size GPU step against the fp64 oracle: the layers are row-separable and the pair
    gradient of a row involves its own group only (the global pair count comes from the labels and groups alone)."""
    R, PO = _oracle()
    ids = np.unique(groups)[:n_groups]
    rows = np.nonzero(np.isin(groups, ids))[0]
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).double() for k, v in named.items()}
    U, V, W, b, K, hk, hb = _split(w)
    x64 = torch.from_numpy(x[rows]).double().requires_grad_(True)
    s64 = R.multi_dense_layer(R.dcn_mix_layer(x64, U, V, W, b, K), hk, hb).reshape(-1)
    if p_total is None:
        _, _, p_total = PO.pairwise_bpr(groups, labels, np.zeros_like(labels), grouped=True)    # pair count: labels and groups only
    lsum, ds, p_sub = PO.pairwise_bpr(groups[rows], labels[rows], s64.detach().numpy().astype(np.float32))
    ds_full = ds * (float(np.float32(p_sub)) + 1e-10) / (float(np.float32(p_total)) + 1e-10)      # same terms, global normalisation
    s64.backward(torch.from_numpy(ds_full))
    return {'rows': int(rows.size), 'pairs_total_cpu': int(p_total), 'pairs_total_gpu': int(n_pair_gpu),
            'scores': rel_err(scores_gpu[rows], s64.detach().numpy()), 'dx': rel_err(dx_gpu[rows], x64.grad.numpy())}


def step_account(lib, run_step, sync, steps=10):
    """Where a step's time goes when launches of two streams overlap: `steps` extra UNTIMED steps with EVERY hooked launch and phase
    recorded (recnow_prof_sample_every(1) + recnow_prof_intervals), then a sweep over the intervals in which every instant is divided
    equally among the launches running at it -- a kernel family's EXCLUSIVE (shared) time per step.  Under one stream this is its plain
    kernel time; under the second stream of the backward pass two co-running products get half of their common time each instead of
    both being charged all of it.  Returns {'per_step_ms': {name: ms}, 'covered_ms_per_step': .., 'launches_per_step': {name: n}}."""
    cap = 512 * steps          # (a product-route step under two streams with phase tags is ~60 records; the model steps of c4 / c5 up to ~200)
    if lib.recnow_prof_enable(cap) != 0 or lib.recnow_prof_sample_every(1) != 0:
        return None
    sync()
    for _ in range(steps):
        run_step()
    sync()
    tags, t0, t1 = (ctypes.c_int * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    n = lib.recnow_prof_intervals(tags, t0, t1, cap)
    dropped = lib.recnow_prof_dropped()          # records that found the pool full: the account would under-report
    lib.recnow_prof_enable(0)
    if n <= 0:
        return None
    names = dict(GEMM_TAGS)
    names.update(PHASE_TAGS)
    ev = []
    for i in range(n):
        if t1[i] > t0[i]:
            ev.append((t0[i], 1, i))
            ev.append((t1[i], 0, i))
    ev.sort(key=lambda e: (e[0], e[1]))          # ends before starts at equal times
    share = [0.0] * n
    active = set()
    last = None
    covered = 0.0
    for t, start, i in ev:
        if active and last is not None and t > last:
            dt = t - last
            covered += dt
            for j in active:
                share[j] += dt / len(active)
        last = t
        if start:
            active.add(i)
        else:
            active.discard(i)
    per, cnt = {}, {}
    for i in range(n):
        k = names.get(tags[i], 'tag %d' % tags[i])
        per[k] = per.get(k, 0.0) + share[i] / steps
        cnt[k] = cnt.get(k, 0) + 1
    return {'per_step_ms': dict(sorted(per.items(), key=lambda kv: -kv[1])), 'covered_ms_per_step': covered / steps,
            'launches_per_step': {k: v / steps for k, v in cnt.items()}, 'truncated_records': int(dropped), 'by_tag': {t: sum(share[i] for i in range(n) if tags[i] == t) / steps for t in set(tags[:n])},
            'what': 'exclusive time per kernel family: %d untimed steps with every hooked launch and phase recorded; an instant shared by k '
                    'running launches counts 1/k for each' % steps}


def self_launch(n_ranks, argv, oversubscribe=False):
    """`python bench.py --gpus N` without a launcher: THIS process has made no GPU call yet (importing torch and counting devices
    do not initialise the runtime) and makes none -- it starts the N ranks as CHILD processes through torch.distributed.run (never an
    exec of a process that touched the GPU), relays their output, prints rank 0's JSON line as the LAST line of stdout and exits
    with the children's return code."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()
    if visible < (1 if oversubscribe else n_ranks):
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible on this host (torch.cuda.device_count() = %d, '
                         'HIP_VISIBLE_DEVICES=%s); one process per GPU, nothing launched'
                         % (n_ranks, visible, visible, os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + [a for a in argv if a != '--launch']
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n_ranks)))
    print('[bench] launching %d ranks: %s' % (n_ranks, ' '.join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            line_json = line
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if line_json is not None:
        sys.stdout.write(line_json)
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
        print('[bench] the ranks exited without a JSON line', file=sys.stderr)
    raise SystemExit(rc)



# ---- configs[3] / configs[4]: data-parallel MODEL steps (VERDICT round 5, row e''): bench.py --config c4 | c5 -------------------------------------------
C5_AUX = (0.5 / 64, 0.5 / 64, -0.5 / 64)        # per-row weights of the pointwise terms of the c5 loss sum (one list = 64 rows: O(1e-2) of the listwise terms)
MODEL_CONFIGS = {
    'c4': {'rows': 16384, 'ranks': 8,
           'workload': 'configs[3]: cin_layer (L=3, H=128) || fm_layer -> Dense(17 -> 1) head -> in-batch pairwise (logistic), 64 fields x 16-dim, '
                       '~8 rows/group; global B = 131072 on 8 GPUs = 16384 rows per GPU'},
    'c5': {'rows': 32768, 'ranks': 8,
           'workload': 'configs[4]: ple_layer (3 tasks + 1 shared group, experts [[512, 256], [256, 128]] x 2) -> 3 heads MultiDense(1, 3) -> '
                       'listwise_loss_from_batch on every task logit, 128 fields x 32-dim = 4096 inputs, 64 rows/list; global B = 262144 on 8 GPUs = 32768 rows per GPU'},
}


def _c4_oracle(named, dtype, F, Dm, n_layers):
    R, _ = _oracle()
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).requires_grad_(True) for k, v in named.items()}

    def fwd(xc):
        fields = [xc[:, f * Dm:(f + 1) * Dm] for f in range(F)]
        feat = torch.cat([R.cin_layer_gemm_form(xc, [w['cin.%d' % k] for k in range(1, n_layers + 1)], F, Dm, True, True), R.fm_layer(fields)], dim=1)
        return R.multi_dense_layer(feat, w['head.kernel'], w['head.bias']).reshape(-1)
    return fwd, w


def _c5_oracle(named, dtype, layer, dims):
    R, _ = _oracle()
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).requires_grad_(True) for k, v in named.items()}
    n_groups = len(layer.task_names[:layer.num_total_task])

    def fwd(xc):          # (rows, 3) task logits
        layers = []
        for l in range(len(dims)):
            entry = {'dnn': [], 'gate': []}
            for gi in range(n_groups):
                scope = 'PLE/ple_layer_%d/task_%s' % (l, layer.task_names[gi])
                entry['dnn'].append([(w['ple.%s/%s/MultiDenseLayer_%d/kernel' % (scope, scope, i)], w['ple.%s/%s/MultiDenseLayer_%d/bias' % (scope, scope, i)])
                                     for i in range(len(dims[l]))])
                gk = 'ple.PLE/ple_gate_%d/task_%s/dense/kernel' % (l, layer.task_names[gi])
                entry['gate'].append((w[gk], w[gk[:-6] + 'bias']) if gk in w else None)
            layers.append(entry)
        outs = R.ple_layer(xc, layers, layer.is_shared_tasks, activation='tanh')
        return R.multi_dense_layer(torch.stack(list(outs)), w['head.kernel'], w['head.bias']).reshape(3, -1).t()
    return fwd, w


def _chunked_fwd(fwd, x, dtype, chunk):
    xt = torch.from_numpy(x)
    outs = []
    with torch.no_grad():
        for lo in range(0, x.shape[0], chunk):
            outs.append(fwd(xt[lo:lo + chunk].to(dtype)).numpy())
    return np.concatenate(outs, 0)


def _chunked_bwd(fwd, x, dout, dtype, chunk):
    xt = torch.from_numpy(x)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    g = torch.from_numpy(np.asarray(dout).astype(npdt))
    dx = np.empty(x.shape, npdt)
    for lo in range(0, x.shape[0], chunk):
        xc = xt[lo:lo + chunk].to(dtype).clone().requires_grad_(True)
        fwd(xc).backward(g[lo:lo + chunk])
        dx[lo:lo + chunk] = xc.grad.numpy()
    return dx


def _listwise_fp(groups, labels, logits, dtype):
    """The reference's dense (G, B) listwise stage on three task logits (+ the pointwise terms C5_AUX): (loss sum, valid lists, d sum / d logits)."""
    R, _ = _oracle()
    lt = torch.from_numpy(np.asarray(logits)).to(dtype).requires_grad_(True)
    total, nv = 0.0, 0
    for t in range(lt.shape[1]):
        _, lab, lg = R.to_listwise_sample(torch.from_numpy(groups), torch.from_numpy(labels).to(dtype), lt[:, t])
        total = total + R.listwise_loss_via_softmax_cross_entropy_with_logits(lab, lg, do_reduce=False).sum()
        nv = int(lab.shape[0])
    for t in range(lt.shape[1]):
        total = total + C5_AUX[t] * lt[:, t].sum()
    total.backward()
    return float(total.item()), nv, lt.grad.numpy()


def main_models(args, world, rank, dev, json_fd, use_dist):
    """`bench.py --config c4 | c5`: the per-rank model step of BASELINE.json's configs[3] / configs[4] through dp.OverlappedGradientReducer (buckets
    all-reduced from autograd's post-accumulate hooks, under the backward pass), the same JSON schema as the headline line, cross-rank fp64 gate."""
    from rec_now_amd import _lib, dp
    lib = _lib.load()
    dp.FORCE_COLLECTIVES = bool(args.force_dist)
    _lib.call('recnow_set_gemm_precision', 0)          # these models' products are the exact-fp32 kernels (no split form of CIN / PLE products)
    cfg = MODEL_CONFIGS[args.config]
    rows = args.rows if args.rows is not None else cfg['rows']
    rng = np.random.default_rng(40 + 1000 * rank)
    torch.manual_seed(5)
    labels = (rng.random(rows) < 0.25).astype(np.float32)
    if args.config == 'c4':
        from rec_now_amd.layers.cin_layer import CINLayer
        from rec_now_amd.layers.fm_layer import FMLayer
        from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
        from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss_fused
        F, Dm, Hs = 64, 16, [128, 128, 128]
        n_groups = max(rows // 8, 2)
        x = rng.normal(0, 0.3, (rows, F * Dm)).astype(np.float32)
        groups = (rng.integers(0, n_groups, rows) + rank * n_groups).astype(np.float32)
        cin, fm, head = CINLayer(Hs), FMLayer(), MultiDenseLayer(1, 1)
        xs = [torch.from_numpy(np.ascontiguousarray(x[:, f * Dm:(f + 1) * Dm])).to(dev).requires_grad_(True) for f in range(F)]
        head(torch.cat([cin([v[:64] for v in xs]), fm([v[:64] for v in xs])], dim=1))          # lazy build
        with torch.no_grad():
            head.kernel.mul_(0.05)
            head.bias.fill_(0.2)
        named = {'cin.%d' % k: cin.idx2weight[k] for k in range(1, len(Hs) + 1)}
        named['head.kernel'], named['head.bias'] = head.kernel, head.bias
        denom = 'eps'
    else:
        from rec_now_amd.layers.multi_dense_layer import MultiDenseLayer
        from rec_now_amd.layers.ple_layer import PLELayer
        from rec_now_amd.rec_block.listwise_loss_from_batch import listwise_loss_from_batch
        Din, dims, n_exp = 4096, [[512, 256], [256, 128]], 2
        n_groups = max(rows // 64, 2)
        x = (rng.normal(0, 0.3, (rows, Din))).astype(np.float32)
        groups = (rng.integers(0, n_groups, rows) + rank * n_groups).astype(np.float32)
        ple, head = PLELayer(3, dims, n_exp, 1, activation='tanh', name='PLE'), MultiDenseLayer(1, 3)
        xs = [torch.from_numpy(x).to(dev).requires_grad_(True)]
        head(torch.stack(list(ple(xs[0][:256]))))
        g = torch.Generator(device='cpu').manual_seed(56)
        with torch.no_grad():
            for name, p in ple.named_weights().items():
                if 'bias' in name:
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).mul_(0.1).to(p.device))
            head.kernel.mul_(6.0)
            head.bias.fill_(0.1)
        named = {'ple.' + k: v for k, v in ple.named_weights().items()}
        named['head.kernel'], named['head.bias'] = head.kernel, head.bias
        denom = 'max1'
    params = list(named.values())
    yd, gd = torch.from_numpy(labels).to(dev), torch.from_numpy(groups).to(dev)
    red = dp.OverlappedGradientReducer(params, denom=denom)
    rank_rows = [rows]
    if use_dist:
        cnt_t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=dev if args.backend == 'nccl' else 'cpu')
        cnt_t[dist.get_rank()] = rows
        dist.all_reduce(cnt_t, op=dist.ReduceOp.SUM)
        rank_rows = [int(v) for v in cnt_t.tolist()]
    total_rows = sum(rank_rows)
    last = {}

    def run_step():
        for p in params:
            p.grad = None
        for v in xs:
            v.grad = None
        if args.config == 'c4':
            scores = head(torch.cat([cin(xs), fm(xs)], dim=1)).reshape(-1)
            local_sum, count = pairwise_loss_fused(scores, yd, gd, reduce_mean=False)
            last['out'] = scores
        else:
            logits = head(torch.stack(list(ple(xs[0])))).reshape(3, -1)
            local_sum, count = 0.0, None
            for t in range(3):
                lmean, nv = listwise_loss_from_batch(gd, yd, logits[t], return_num_list=True)
                local_sum = local_sum + lmean * nv.detach()
                count = nv.detach().to(torch.float32)
            # + small pointwise terms on the task logits (as tests/test_northstar_gpu.py::test_config5_end_to_end..: the listwise gradient sums to zero inside
            # every list, so bias gradients made of it alone are sums that cancel and a relative bound on them measures the conditioning of the sum)
            local_sum = local_sum + C5_AUX[0] * logits[0].sum() + C5_AUX[1] * logits[1].sum() + C5_AUX[2] * logits[2].sum()
            last['out'] = logits
        red.prepare(local_sum, count)
        local_sum.backward()
        loss, cnt = red.finish()
        last['count'] = cnt
        return loss

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        run_step()
    prof = not args.no_prof
    if prof:
        _lib.check(lib.recnow_prof_enable(512 * (args.steps + 1)), 'recnow_prof_enable')
        _lib.check(lib.recnow_prof_sample_every(1), 'recnow_prof_sample_every')
    import gc
    gc.collect()
    gc.disable()
    for _ in range(2):
        run_step()
    if prof:
        _c, _m, _f, _b = (ctypes.c_int * N_TAGS)(), (ctypes.c_double * N_TAGS)(), (ctypes.c_double * N_TAGS)(), (ctypes.c_double * N_TAGS)()
        _lib.check(lib.recnow_prof_collect(_c, _m, _f, _b), 'recnow_prof_collect')
        _lib.check(lib.recnow_prof_sample_every(1), 'recnow_prof_sample_every')
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    roofline = None
    if prof:
        cnt, ms, fl, by = (ctypes.c_int * N_TAGS)(), (ctypes.c_double * N_TAGS)(), (ctypes.c_double * N_TAGS)(), (ctypes.c_double * N_TAGS)()
        _lib.check(lib.recnow_prof_collect(cnt, ms, fl, by), 'recnow_prof_collect')
        lib.recnow_prof_enable(0)
    account = step_account(lib, run_step, sync, steps=4) if prof else None
    comm = None
    if use_dist:
        def timed_ms(n=6):
            sync()
            c0 = time.perf_counter()
            for _ in range(n):
                run_step()
            sync()
            tt = torch.tensor([(time.perf_counter() - c0) * 1e3 / n], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        with_ms = timed_ms()
        dp._SKIP_COLLECTIVE = True
        try:
            without_ms = timed_ms()
        finally:
            dp._SKIP_COLLECTIVE = False
        comm = {'with_collectives_ms': with_ms, 'without_collectives_ms': without_ms, 'comm_exposed_ms': with_ms - without_ms,
                'buckets': [int(f.numel() * 4) for f in red._flat],
                'what': 'ms per step over 6 untimed steps (max over the ranks) with and without the bucket all-reduces; buckets (bytes) in the order they '
                        'are reduced, each from the autograd hook that completes it'}
    if prof:
        tag = max(GEMM_TAGS, key=lambda t: ms[t])
        if account is not None:
            excl = {t: account['by_tag'].get(t, 0.0) for t in GEMM_TAGS if cnt[t] > 0}
            if excl:
                tag = max(excl, key=lambda t: excl[t])
        if cnt[tag] > 0 and ms[tag] > 0:
            achieved = fl[tag] / (ms[tag] * 1e-3) / 1e12
            roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS,
                        'traffic': None, 'kernel': GEMM_TAGS[tag], 'launches': cnt[tag], 'sampled': 'every hooked launch of the timed region',
                        'avg_launch_us': ms[tag] * 1e3 / cnt[tag], 'algorithmic_flops_per_launch': fl[tag] / cnt[tag],
                        'all_gemm': {GEMM_TAGS[t]: {'launches': cnt[t], 'ms': ms[t], 'tflops': (fl[t] / (ms[t] * 1e-3) / 1e12) if ms[t] > 0 else None}
                                     for t in GEMM_TAGS if cnt[t] > 0},
                        'whole_step_tflops': sum(fl[t] for t in GEMM_TAGS) / args.steps / (elapsed / args.steps) / 1e12}
            if account is not None:
                roofline['exclusive_ms_per_step'] = account['per_step_ms']
                roofline['exclusive_covered_ms_per_step'] = account['covered_ms_per_step']
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        if args.backend != 'nccl':
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- the gate (untimed): one more step, every rank holds its rows to the fp64 oracle, the loss stage and the reduced gradients to the gathered batch
    R, PO = _oracle()
    loss_q = run_step()
    torch.cuda.synchronize()
    named_np = {k: v.detach().cpu().numpy() for k, v in named.items()}
    out_gpu = last['out'].detach().cpu().numpy()
    dx_gpu = (torch.cat([v.grad for v in xs], dim=1) if args.config == 'c4' else xs[0].grad).cpu().numpy()
    count_glob = float(last['count'].item())
    dx_gpu = dx_gpu / (np.float32(count_glob) + np.float32(1e-10) if denom == 'eps' else max(count_glob, 1.0))      # x is not a parameter: the loss SUM's gradient
    full_gate = not args.no_cpu_baseline
    world_n = dist.get_world_size() if use_dist else 1

    def gather(a):
        if not use_dist:
            return np.ascontiguousarray(a)
        a = np.ascontiguousarray(a)
        tt = torch.from_numpy(a)
        if args.backend == 'nccl':
            tt = tt.to(dev)
        parts = [torch.empty_like(tt) for _ in range(world_n)]
        dist.all_gather(parts, tt)
        return torch.cat(parts).cpu().numpy()

    def allsum64(v):
        tt = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        if use_dist:
            if args.backend == 'nccl':
                tt = tt.to(dev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        return tt.cpu().numpy()
    chunk = 512 if args.config == 'c4' else 2048
    parity = {'tolerance': PARITY_TOL, 'loss': float(loss_q.item()), 'ranks': world_n}
    worst = 0.0
    if args.config == 'c4':
        g_all, y_all, s_all = gather(groups), gather(labels), gather(out_gpu)
        o_loss, _, o_pairs = PO.pairwise_bpr(g_all, y_all, s_all, grouped=True)
        parity['gathered_batch_pairs_oracle'] = {'loss': rel_err(float(loss_q.item()), o_loss), 'pairs_gpu': int(count_glob), 'pairs_oracle': int(o_pairs)}
        worst = parity['gathered_batch_pairs_oracle']['loss']
        ok = int(count_glob) == int(o_pairs)
        if full_gate:
            fwd, wl = _c4_oracle(named_np, torch.float64, 64, 16, 3)
            s64 = _chunked_fwd(fwd, x, torch.float64, chunk)
            s64_all = gather(s64)
            f_loss, ds_all, f_pairs = PO.pairwise_bpr(g_all, y_all, s64_all, grouped=True)
            lo = sum(rank_rows[:rank])
            dx64 = _chunked_bwd(fwd, x, ds_all[lo:lo + rows], torch.float64, chunk)
    else:
        if full_gate:
            fwd, wl = _c5_oracle(named_np, torch.float64, ple, [[512, 256], [256, 128]])
            l64 = _chunked_fwd(fwd, x, torch.float64, chunk)
        else:
            l64 = out_gpu.T.astype(np.float64)
        # the listwise stage on the GPU's own logits (fp64, dense (G, B) form of the reference): lists live on one rank, so the global mean is sum / sum
        lsum_g, nv_l, _ = _listwise_fp(groups, labels, out_gpu.T.astype(np.float64), torch.float64)
        tot = allsum64(np.array([lsum_g, nv_l]))
        parity['listwise_stage_oracle'] = {'loss': rel_err(float(loss_q.item()), tot[0] / max(tot[1], 1.0)), 'lists_gpu': int(count_glob), 'lists_oracle': int(tot[1])}
        worst = parity['listwise_stage_oracle']['loss']
        ok = int(count_glob) == int(tot[1])
        if full_gate:
            lsum64, nv64, dl64 = _listwise_fp(groups, labels, l64, torch.float64)
            tot64 = allsum64(np.array([lsum64, nv64]))
            f_loss = tot64[0] / max(tot64[1], 1.0)
            dx64 = _chunked_bwd(fwd, x, dl64 / max(tot64[1], 1.0), torch.float64, chunk)
            s64 = l64
    if full_gate:
        names = sorted(named)
        flat = allsum64(np.concatenate([(wl[k].grad.numpy() if wl[k].grad is not None else np.zeros(tuple(wl[k].shape))).reshape(-1) for k in names]))
        full = {'loss': rel_err(float(loss_q.item()), f_loss), 'outputs': rel_err(out_gpu if args.config == 'c4' else out_gpu.T, s64), 'dx': rel_err(dx_gpu, dx64)}
        loc = torch.tensor([full['outputs'], full['dx']], dtype=torch.float64, device=dev)
        if use_dist:
            loc = loc if args.backend == 'nccl' else loc.cpu()
            dist.all_reduce(loc, op=dist.ReduceOp.MAX)
        full['outputs'], full['dx'] = float(loc[0].item()), float(loc[1].item())
        off = 0
        gsum = float(np.abs(ds_all).sum()) if args.config == 'c4' else None
        for k in names:
            n = named[k].numel()
            ref = flat[off:off + n].reshape(tuple(named[k].shape))
            off += n
            # a gradient that is a sum cancelling to ~0 (the head bias under a pairwise loss): on the scale of its terms
            full[k] = rel_err(named[k].grad.cpu().numpy(), ref, scale=gsum if (k == 'head.bias' and args.config == 'c4') else None)
        parity['oracle_fp64_all_ranks'] = full
        worst = max([worst] + list(full.values()))
    parity['parity_max_rel'] = worst
    parity['ok'] = bool(ok and worst <= PARITY_TOL)

    # ---- CPU baseline (rank 0 at N = 1 only): the fp32 port on a bounded sample of the same workload
    cpu = None
    if full_gate and world == 1 and not use_dist:
        n_s = min(rows, 2048 if args.config == 'c4' else 8192)
        ids = np.unique(groups)
        sel = np.nonzero(np.isin(groups, ids[:max(1, int(len(ids) * n_s / rows))]))[0]          # whole groups, ~n_s rows
        xs_, gs_, ys_ = np.ascontiguousarray(x[sel]), groups[sel], labels[sel]
        c0 = time.perf_counter()
        if args.config == 'c4':
            fwd32, _w = _c4_oracle(named_np, torch.float32, 64, 16, 3)
            sc = _chunked_fwd(fwd32, xs_, torch.float32, chunk)
            _, ds, _ = PO.pairwise_bpr(gs_, ys_, sc, grouped=True)
            _chunked_bwd(fwd32, xs_, ds, torch.float32, chunk)
        else:
            fwd32, _w = _c5_oracle(named_np, torch.float32, ple, [[512, 256], [256, 128]])
            lg = _chunked_fwd(fwd32, xs_, torch.float32, chunk)
            _, nvs, dl = _listwise_fp(gs_, ys_, lg, torch.float32)
            _chunked_bwd(fwd32, xs_, dl / max(nvs, 1), torch.float32, chunk)
        c_sec = time.perf_counter() - c0
        cpu = {'value': len(sel) / c_sec, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': '1 step of fwd+bwd on %d rows of whole groups of the same batch (%.1f s): oracle/dense_ref.py layers in %d-row chunks on torch-CPU fp32 + '
                         'the loss stage of the oracle; TF2 itself is not installable here' % (len(sel), c_sec, chunk)}

    if rank == 0:
        out = {'metric': 'samples/sec fwd+bwd, %s' % ('cin_layer + fm_layer + in-batch pairwise, data-parallel' if args.config == 'c4' else 'ple_layer 3-task + listwise_loss_from_batch, data-parallel'),
               'value': total_rows * args.steps / elapsed, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': cfg['workload'] + '; this run: %d rows per GPU on %d GPU(s)' % (rows, world), 'rows_per_gpu': rows, 'rows_per_rank': rank_rows,
                          'global_batch': total_rows, 'parallelism': 'dp%d' % world, 'input_grad': True,
                          'route': 'drop-in layers through autograd; gradients all-reduced per bucket from post-accumulate hooks under the backward pass '
                                   '(dp.OverlappedGradientReducer), loss statistics in the first bucket',
                          'loss': float(loss.item())},
               'roofline': roofline, 'parity': parity, 'parity_max_rel': parity['parity_max_rel']}
        if cpu is not None:
            out['cpu_baseline'] = cpu
        if use_dist:
            out['rccl_ranks'] = world_n
            out['backend'] = args.backend
            if comm is not None:
                out['comm_exposed_ms'] = comm['comm_exposed_ms']
                out['comm'] = comm
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--config', choices=['c3', 'c4', 'c5'], default='c3', help="c3 (default): BASELINE.json's metric (configs[2], the headline); c4 / c5: the "
                    'data-parallel MODEL steps of configs[3] / configs[4] at their per-rank size (16 384 / 32 768 rows per GPU; --rows overrides), through dp.OverlappedGradientReducer')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='do not record per-launch HIP events in the timed region')
    ap.add_argument('--graph', action='store_true', help='replay the step from captured HIP graphs (SURVEY 8f.1; one graph per phase / backward piece) '
                    'instead of enqueueing it eagerly')
    ap.add_argument('--eager', action='store_true', help='never replay graphs (overrides --graph)')
    ap.add_argument('--route', choices=['step', 'autograd'], default='step',
                    help="'step' (default): the whole step through recnow_dcn_mix_step (rec_now_amd/step.py: one C call per phase, buffers allocated "
                         "once); 'autograd': the model-level fused node dcn_mix_score + pairwise_loss_fused through torch.autograd")
    ap.add_argument('--no-input-grad', action='store_true', help='diagnostic: x is data without a gradient (the metric keeps d loss / d x: in a model x is the embedding output)')
    ap.add_argument('--scaling', choices=['strong', 'weak'], default='strong',
                    help="'strong' (default, the metric: B = 65536 is the GLOBAL batch, every rank owns 65536 / N rows of whole groups); "
                         "'weak': 65536 rows per GPU")
    ap.add_argument('--rows', type=int, default=None, help='rows per GPU (diagnostics: the per-rank shard of the 2/4/8-GPU rows on one GPU, '
                    'e.g. --rows 8192 --force-dist; default 65536 / N under strong scaling, 65536 under weak scaling)')
    ap.add_argument('--shard', choices=['blocks', 'hash'], default='blocks',
                    help="how the global batch reaches the ranks.  'blocks' (default): every rank draws its own rows-per-GPU rows of whole groups with ids "
                         "distinct across ranks (equal, 256-aligned shards).  'hash': ONE global batch (the same on every rank) split by "
                         "rec_now_amd.dp.shard_rows_by_group -- every group on one rank, i.e. RAGGED per-rank batches (B %% 256 != 0), which the step runs on "
                         "padded storage (recnow_dcn_mix_step_desc.B_pad); the cross-rank gate is the same")
    ap.add_argument('--group-inline', action='store_true', help='diagnostic: the grouping of the batch on the main stream instead of a side stream under the forward pass')
    ap.add_argument('--hostprof', default=None, help='diagnostic: cProfile the host side of 20 extra (untimed) steps into this file')
    ap.add_argument('--unfused', action='store_true', help='diagnostic: the drop-in composition head(cross(x)) and pairwise_loss(outputs, labels, groups) '
                    'instead of the model-level fused node (rec_now_amd/fused.py) with grouping on a side stream')
    ap.add_argument('--gemm-precision', choices=['auto', 'f32', 'bf16x3'], default='auto',
                    help="arithmetic of the 18 products of the step: 'f32' exact fp32 MFMA; 'bf16x3' the split-precision kernels (fp32 operands as three "
                         "bf16 pieces, six bf16 MFMA terms, fp32 accumulation: per-product error <= 2^-25, same 1e-5 parity gate, not bit-identical); "
                         "'auto' (default): whichever is faster at this shard size under that gate -- bf16x3 from 32 768 rows per GPU (product route), "
                         "f32 below (row-block kernels).  The other mode's step time is measured in the same run and reported beside the headline "
                         "(`exact_f32` / `split_precision` at the end of the JSON line)")
    ap.add_argument('--no-other', action='store_true', help='do not time the other arithmetic of the products behind the headline (profiling runs: one kernel family per trace)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the RCCL process group even at world size 1 (exercises the N>1 code path on one GPU)')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl', help="process-group backend: 'nccl' (= RCCL over xGMI, the product path); 'gloo' is a "
                    'diagnostic that lets several ranks share one GPU (with --oversubscribe), so that the N > 1 step, reducer and cross-rank gate run with real '
                    'multi-rank semantics on a one-GPU box (tests/test_bench_gpu.py)')
    ap.add_argument('--oversubscribe', action='store_true', help='diagnostic (gloo only): rank r uses GPU r %% (visible GPUs) instead of requiring one GPU per rank')
    ap.add_argument('--launch', action='store_true', help='start the ranks through the self-launcher even when --gpus is 1 (tests: --gpus 1 --force-dist --launch); '
                    '--gpus N > 1 without WORLD_SIZE in the environment always takes it')
    args = ap.parse_args()
    if args.oversubscribe and args.backend != 'gloo':
        raise SystemExit('--oversubscribe needs --backend gloo (RCCL refuses two ranks on one device)')
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.launch):
        self_launch(args.gpus, sys.argv[1:], oversubscribe=args.oversubscribe)          # before ANY device call; does not return

    # The N > 1 step uses four streams (main, grouping, gradient all-reduce, RCCL's own).  With HIP's default of 4 hardware queues
    # two of them shared a queue: either the grouping ran in front of the forward pass instead of under it, or the all-reduce
    # stream sat behind the whole backward pass (kernel trace, +0.09 ms per step).  The HIP runtime reads this when it
    # initialises, i.e. at the first device call below.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes on this driver); also set by self_launch, here for a rank started by an outside launcher
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # stdout carries ONE line, the JSON of rank 0: from here on file descriptor 1 points at stderr, so whatever a library prints through C stdio or
    # Python's sys.stdout during the run (RCCL writes a five-line version banner to stdout when its first communicator comes up) lands there; the JSON
    # line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if args.oversubscribe:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')

    # host legs (CPU baseline, fp64 gate): threads from the cgroup's CPU share, not from the 256 logical CPUs a GPU box shows
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 2 * cpu_share()) // max(world, 1)))
    from rec_now_amd import _lib as _l0
    assert _l0.load().recnow_prof_tag_count() <= N_TAGS, 'recnow_prof_collect fills more tags than bench.py has room for'
    if args.config != 'c3':
        return main_models(args, world, rank, dev, json_fd, use_dist)
    from rec_now_amd import _lib, dp
    from rec_now_amd.rec_block.pairwise_loss_from_batch import group_rows, pairwise_loss_fused
    lib = _lib.load()
    dp.FORCE_COLLECTIVES = bool(args.force_dist)

    torch.manual_seed(3)                      # identical replicated weights on every rank
    model = Model()
    if args.rows is None:
        if args.scaling == 'strong' and GLOBAL_BATCH % world:
            raise SystemExit('strong scaling shards B = %d over %d ranks: not divisible' % (GLOBAL_BATCH, world))
        rows = GLOBAL_BATCH // world if args.scaling == 'strong' else B_PER_GPU
    else:
        rows = args.rows
    if args.shard == 'hash':
        # ONE global batch, identical on every rank; rank r keeps the rows whose group hashes to r (whole groups, ragged shard sizes)
        xg, gg, yg = synth_batch(rows * world, 3, 0)
        mine = np.nonzero(dp.shard_rows_by_group(gg.astype(np.int64), world).numpy() == rank)[0]
        x, groups, labels = np.ascontiguousarray(xg[mine]), np.ascontiguousarray(gg[mine]), np.ascontiguousarray(yg[mine])
        rows = int(mine.size)
        del xg, gg, yg
    else:
        x, groups, labels = synth_batch(rows, 3, rank)
    # the arithmetic of the products (see --gemm-precision): chosen from the NOMINAL shard size, so that every rank takes the same mode
    nominal_rows = rows if args.shard != 'hash' else (args.rows if args.rows is not None else GLOBAL_BATCH // world)
    precision = args.gemm_precision if args.gemm_precision != 'auto' else ('bf16x3' if nominal_rows >= SPLIT_FROM_ROWS else 'f32')
    _lib.call('recnow_set_gemm_precision', 1 if precision == 'bf16x3' else 0)
    # rows of every rank (ragged under --shard hash): the whole-job throughput counts all of them, the cross-rank gate slices by them
    rank_rows = [rows]
    if use_dist:
        cnt_t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=dev if args.backend == 'nccl' else 'cpu')
        cnt_t[dist.get_rank()] = rows
        dist.all_reduce(cnt_t, op=dist.ReduceOp.SUM)
        rank_rows = [int(v) for v in cnt_t.tolist()]
    total_rows = sum(rank_rows)
    xd, gd, yd = (torch.from_numpy(v).to(dev) for v in (x, groups, labels))
    model(xd[:256])                           # lazy build on the device
    # The step includes the gradient w.r.t. x (in a model x is the concatenated embedding output and needs it); a data-only
    # x lets DCNMixLayer drop every dx product (--no-input-grad, reported as a diagnostic only).
    xd.requires_grad_(not args.no_input_grad)
    params = [p for p in model.parameters()]
    reducer = dp.GradientAllReducer(params)
    # fused route (default): cross layers + scoring head as one node; per-layer events let the gradient all-reduce of a layer run
    # under the backward of the layers below it (dp.LayerwiseReducer).  Stages in the order their gradients become final.
    from rec_now_amd.fused import GpuEvent, dcn_mix_score, fused_route_available
    from rec_now_amd.rec_block.pairwise_loss_from_batch import pairwise_loss
    fused = not args.unfused and fused_route_available(model.cross, model.head, xd)
    use_step = fused and args.route == 'step'
    # under a process group the step stays EAGER by default: the whole-step entry leaves the host at 0.3-0.4 ms per step, below the GPU
    # time of even the 8192-row shard, and an eager backward pass can run its weight-gradient products on a second stream across the
    # layers (measured: 0.82 vs 0.83 ms replayed at 8192 rows, 1.22 vs 1.27 at 16 384); --graph replays per-piece graphs instead
    use_graph = args.graph and not args.eager
    two_streams = os.environ.get('RECNOW_STEP_TWO_STREAMS', '1' if (use_dist and not use_graph) else '0') == '1'
    events, layerwise, grad_buffers, pstep = None, None, None, None
    if use_step:
        # whole-step entry: one C call per phase on buffers allocated once; under a process group the gradients are produced inside the
        # reducer's per-layer buckets and every bucket's all-reduce is enqueued as soon as its piece of the backward pass is
        from rec_now_amd.step import DCNMixPairwiseStep
        if use_dist:
            stages = DCNMixPairwiseStep.stages_for(model.cross, model.head)
            layerwise = dp.LayerwiseReducer(stages, [GpuEvent() for _ in stages], dev)
        pstep = DCNMixPairwiseStep(model.cross, model.head, xd.detach(), yd, gd, need_dx=not args.no_input_grad, reducer=layerwise,
                                   two_streams=two_streams)
    elif fused and use_dist:
        events = [GpuEvent() for _ in range(LAYERS)]
        per_layer = lambda l: [model.cross.origin_to_sub_kernels[l], model.cross.sub_to_sub_kernels[l], model.cross.sub_to_origin_kernels[l],     # noqa: E731
                               model.cross.biases[l], model.cross.gate_layers[l].kernel]
        stages = [per_layer(LAYERS - 1) + [model.head.kernel, model.head.bias]] + [per_layer(l) for l in range(LAYERS - 2, -1, -1)]
        layerwise = dp.LayerwiseReducer(stages, [events[l] for l in range(LAYERS - 1, -1, -1)], dev)
        from rec_now_amd.fused import score_params
        grad_buffers = [layerwise.buffer_of(p) for p in score_params(model.cross, model.head)]     # gradients are written into the buckets

    side = torch.cuda.Stream(device=dev)
    last = {}

    def step(xin=None):
        if pstep is not None:
            if xin is not None:
                pstep.x.copy_(xin.detach())       # the parity step's inputs (the step object owns its input storage)
            loss_val, n_pair = pstep.run()
            last['scores'], last['n_pair'], last['dx'] = pstep.scores, n_pair, pstep.dx
            return loss_val
        xin = xd if xin is None else xin
        for p in params:
            p.grad = None
        xin.grad = None
        # the grouping of the batch (sort by group id, segments) does not depend on the scores: it runs on a side stream
        # under the forward pass.  Still part of the step: the group ids are an input of every step.
        if args.unfused:
            # the drop-in composition exactly as a user of the reference writes it
            scores = model(xin)
            if not use_dist:
                loss_val, n_pair = pairwise_loss(scores, yd, gd, return_num_pair=True)
                loss_val.backward()
                last['scores'], last['n_pair'] = scores, n_pair
                return loss_val.detach()
            local_sum, n_pair = pairwise_loss_fused(scores, yd, gd, reduce_mean=False)
        else:
            if args.group_inline:
                seg = group_rows(gd)
                scores = dcn_mix_score(model.cross, model.head, xin, layer_events=events, grad_buffers=grad_buffers) if fused else model(xin)
            else:
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    seg = group_rows(gd)
                scores = dcn_mix_score(model.cross, model.head, xin, layer_events=events, grad_buffers=grad_buffers) if fused else model(xin)
                main.wait_stream(side)
            if not use_dist:
                # one process: the reference's own normalisation (pairwise_loss_from_batch.py:279, mean over the pairs) inside the
                # loss kernel, as `pairwise_loss` returns it -- no statistics to combine
                loss_val, n_pair = pairwise_loss_fused(scores, yd, gd, reduce_mean=True, segments=seg)
                loss_val.backward()
                last['scores'], last['n_pair'] = scores, n_pair
                return loss_val.detach()
            local_sum, n_pair = pairwise_loss_fused(scores, yd, gd, reduce_mean=False, segments=seg)
        if layerwise is not None:
            # backward on the unnormalised local sum; each layer's bucket is all-reduced behind its event while the layers below
            # are still in their backward pass; (loss sum, P) ride in the first bucket
            layerwise.prepare(local_sum, n_pair)
            local_sum.backward()
            loss_val, _ = layerwise.reduce(local_sum, n_pair)
        elif use_dist:
            # one collective per step: backward on the unnormalised local sum; (loss sum, P) ride in the gradient bucket and
            # the gradients are divided by P_global afterwards (the loss is linear in 1/P) -- no sync between fwd and bwd
            local_sum.backward()
            loss_val, _ = reducer.all_reduce_with_loss(local_sum, n_pair)
        else:                 # N > 1 without the gradient reducers (not reached by the flags of this script; kept as the two-collective form)
            loss_bw, loss_val, _ = dp.global_pairwise_loss(local_sum, n_pair)
            loss_bw.backward()
        last['scores'], last['n_pair'] = scores, n_pair
        return loss_val

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    trace_on = os.environ.get('RECNOW_BENCH_TRACE') in ('1', '2')

    def mark(what):          # diagnostic: phase markers on stderr (RECNOW_BENCH_TRACE=1: behind a device synchronisation; =2: host progress only)
        if trace_on:
            if os.environ.get('RECNOW_BENCH_TRACE') == '1':
                torch.cuda.synchronize()
            print('[bench] %s' % what, file=sys.stderr, flush=True)
    if trace_on and pstep is not None:
        print('[bench] ws %x +%d  x %x  dx %x  scores %x  buckets %s' % (pstep.ws.data_ptr(), pstep.ws.numel(), pstep.x.data_ptr(), pstep.dx.data_ptr() if pstep.dx is not None else 0,
              pstep.scores.data_ptr(), [(hex(f.data_ptr()), f.numel() * 4) for f in (layerwise._flat if layerwise is not None else [])]), file=sys.stderr, flush=True)

    graph = None
    hook_pre = None
    if pstep is not None:
        for _ in range(max(args.warmup, 1 if use_graph else 0)):
            step()
        mark('eager warm-up done')
        if use_graph and not args.no_prof:
            # the event hook cannot bracket the kernel nodes of a replayed graph: the roofline samples of a graph run come from 10 more
            # EAGER steps here, before anything is captured -- same kernels, same launch order, same buffers as the timed replays
            _lib.check(lib.recnow_prof_enable(64 * 12), 'recnow_prof_enable')
            _lib.check(lib.recnow_prof_sample_every(PROF_EVERY), 'recnow_prof_sample_every')
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            hook_pre = tuple((ctypes.c_int * N_TAGS)() if i == 0 else (ctypes.c_double * N_TAGS)() for i in range(4))
            _lib.check(lib.recnow_prof_collect(*hook_pre), 'recnow_prof_collect')
            lib.recnow_prof_enable(0)
            mark('eager hook steps done')
        if use_graph:
            pstep.capture()
            mark('captured')
            graph = pstep
            run_step = lambda: pstep.replay()[0]      # noqa: E731
            for _ in range(2):
                run_step()
                mark('replayed once')
    elif not use_graph:
        for _ in range(args.warmup):
            step()
    else:
        if use_dist:
            raise SystemExit('--graph with --route autograd / --unfused is a single-GPU diagnostic')
        # every warm-up step on a non-default stream (torch.cuda.graph's own recipe), and no reference to an earlier step's autograd graph
        # left: the AccumulateGrad nodes of the parameters then belong to a capturable stream -- created under the legacy default stream
        # (and kept alive through last['scores']) they make hipStreamEndCapture fail
        cap_warm = torch.cuda.Stream()
        cap_warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap_warm):
            for _ in range(max(args.warmup, 3)):
                step()
        torch.cuda.current_stream().wait_stream(cap_warm)
        torch.cuda.synchronize()
        last.clear()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_loss = step()
        run_step = lambda: (graph.replay(), graph_loss)[1]     # noqa: E731
        for _ in range(2):
            run_step()
    if graph is None:
        run_step = step
    prof = not args.no_prof and graph is None
    # the two timing events around a sampled launch serialise the stream for ~2 us each: nothing at 65 536 rows (24 hooked launches of ~150 us per
    # step), but 0.03 ms of a 0.64 ms step at 8192 rows when every 5th launch is sampled (measured: 0.640 timed vs 0.611 ms unhooked).  The
    # shards sample every 7th hooked launch instead (about one per step at the 8 hooked launches of a row-block step, four at the 27 of the
    # product route; 7 is coprime to both, so every launch position comes up within a few steps: with every 23rd the backward chain's
    # position went unsampled in a 20-step run and the line named the forward launch).
    prof_every = PROF_EVERY if rows >= 65536 else 7
    if prof:
        _lib.check(lib.recnow_prof_enable(64 * (args.steps + 1)), 'recnow_prof_enable')
        _lib.check(lib.recnow_prof_sample_every(prof_every), 'recnow_prof_sample_every')
    # The cyclic garbage collector stays out of the timed region: a full (generation-2) collection of this process's heap takes 35-65 ms -- seen in
    # round 5 as ONE step of 36-66 ms at a fixed wall time after start-up (whichever step that was), followed by five slower steps while the GPU's
    # clocks recovered from the idle gap; the autograd routes, which allocate graph objects every step, hit it inside 20 timed steps (5.2 ms
    # average against 3.4), the step entry allocates nothing and never did.  Collected once here, disabled until the timed steps are done.
    import gc
    gc.collect()
    gc.disable()
    # ... and the GPU has idled for tens of milliseconds under that collection and the creation of the hook's events: its clocks take ~5 steps to
    # come back (per-step times after such a gap: 4.3 4.2 3.9 3.7 3.65 ... 3.45 ms; with eight more untimed steps still 3.58 3.51 3.44 3.36 3.35 3.31
    # ... 3.24).  60 ms of further UNTIMED steps close the gap; their hook samples are discarded, so the roofline samples are launches of the timed
    # steps only.  (`config.untimed_steps` says how many steps ran before the timed ones in all.)
    # The ramp is a matter of time, not of steps -- but the COUNT must be the same on every rank (a step holds collectives: ranks that ran different
    # numbers of steps would wait for each other forever), so it is derived from the shard size, not from a clock: ~60 ms of steps at the
    # step time this shape is known to take (3.3 ms per 65 536 rows + 0.25 ms of fixed cost), at least three.
    rewarm = max(REWARM_STEPS, min(120, int(REWARM_SECONDS * 1e3 / (3.3 * max(rank_rows) / 65536.0 + 0.25)) + 1))      # (rank_rows: identical on every rank)
    for _ in range(rewarm):
        run_step()
    if prof:
        _c, _m, _f, _b = (ctypes.c_int * N_TAGS)(), (ctypes.c_double * N_TAGS)(), (ctypes.c_double * N_TAGS)(), (ctypes.c_double * N_TAGS)()
        _lib.check(lib.recnow_prof_collect(_c, _m, _f, _b), 'recnow_prof_collect')      # (synchronises; rewinds the sample pool)
        _lib.check(lib.recnow_prof_sample_every(prof_every), 'recnow_prof_sample_every')
    sync()
    if os.environ.get('RECNOW_BENCH_STEPTIMES') == '1':      # diagnostic: every step's own wall time, device-synchronised (changes what is measured)
        per = []
        for _ in range(args.steps):
            c0 = time.perf_counter()
            run_step()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - c0) * 1e3)
        print('[bench] per-step ms: %s' % ' '.join('%.2f' % v for v in per), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    mark('timed region done')
    roofline = None
    prof_note = 'every %dth hooked launch of the timed region' % (PROF_EVERY if rows >= 65536 else 7)
    hook_done = False
    if not prof and hook_pre is not None:
        prof, hook_done = True, True
        prof_note = 'every %dth hooked launch of 10 EAGER steps run before the graphs were captured (the timed steps replay HIP graphs)' % PROF_EVERY
        cnt, ms, fl, by = hook_pre
    if prof and not hook_done:               # collected and switched off HERE: the samples are launches of the timed steps only
        cnt = (ctypes.c_int * N_TAGS)()          # the library fills RN_TAG_MAX (= 12) entries
        ms = (ctypes.c_double * N_TAGS)()
        fl = (ctypes.c_double * N_TAGS)()
        by = (ctypes.c_double * N_TAGS)()
        _lib.check(lib.recnow_prof_collect(cnt, ms, fl, by), 'recnow_prof_collect')
        lib.recnow_prof_enable(0)
    # diagnostic: host time to ENQUEUE a step (no synchronisation inside): well below ms_per_step = the step is GPU-bound
    h0 = time.perf_counter()
    for _ in range(5):
        run_step()
    host_ms = (time.perf_counter() - h0) * 1e3 / 5
    sync()
    mark('host enqueue loop done')
    if args.hostprof:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            run_step()
        pr.disable()
        sync()
        with open(args.hostprof, 'w') as fh:
            pstats.Stats(pr, stream=fh).sort_stats('cumulative').print_stats(45)
    # diagnostics of the step route (untimed, after the timed region): the exclusive-time account and, under a process group, what the
    # collectives add to a step (the same step with RECNOW_DP_SKIP_COLLECTIVE semantics switched on for ten steps on every rank)
    account, comm = None, None
    if pstep is not None and graph is None and not args.no_prof:
        account = step_account(lib, run_step, sync)
        mark('step account done')
    if use_dist and pstep is not None and graph is None:
        def timed_ms(n=10):
            sync()
            c0 = time.perf_counter()
            for _ in range(n):
                run_step()
            sync()
            tt = torch.tensor([(time.perf_counter() - c0) * 1e3 / n], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        with_ms = timed_ms()
        dp._SKIP_COLLECTIVE = True
        try:
            without_ms = timed_ms()
        finally:
            dp._SKIP_COLLECTIVE = False
        comm = {'with_collectives_ms': with_ms, 'without_collectives_ms': without_ms, 'comm_exposed_ms': with_ms - without_ms,
                'what': 'ms per step over 10 untimed steps (max over the ranks), with and without the gradient all-reduces (event hops and the '
                        'scale launches stay): the difference is what the collectives cost a step after the overlap with the backward pass'}
        mark('comm probe done')
    if prof:
        traffic_tab, stale = {}, None
        try:
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as fh:
                tj = json.load(fh)
            stale = tj.get('kernel_source_sha256') != kernel_source_hash()
            if not stale:
                traffic_tab = tj['hbm_bytes_per_launch']
        except (OSError, ValueError, KeyError):
            stale = None                      # no PMC table at all
        tag = max(GEMM_TAGS, key=lambda t: ms[t])
        if account is not None:
            # the kernel family with the largest EXCLUSIVE time (launches of the second stream are timed while they overlap: their summed
            # event times count the shared time twice)
            excl = {t: account['by_tag'].get(t, 0.0) for t in GEMM_TAGS if cnt[t] > 0}
            if excl:
                tag = max(excl, key=lambda t: excl[t])
        if cnt[tag] > 0:
            achieved = fl[tag] / (ms[tag] * 1e-3) / 1e12
            # k_gemm_split executes every algorithmic fp32 multiply-add as six bf16 MFMA terms: its ceiling is the dense bf16 peak / 6
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if tag == 8 else PEAK_F32_MFMA_TFLOPS
            roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': achieved / peak, 'traffic': traffic_tab.get(GEMM_TAGS[tag]), 'traffic_stale': stale,
                        'kernel': GEMM_TAGS[tag],
                        'launches': cnt[tag], 'sampled': prof_note, 'avg_launch_us': ms[tag] * 1e3 / cnt[tag],
                        'algorithmic_flops_per_launch': fl[tag] / cnt[tag], 'algorithmic_bytes_per_launch': by[tag] / cnt[tag],
                        'all_gemm': {GEMM_TAGS[t]: {'launches': cnt[t], 'ms': ms[t],
                                                    'tflops': (fl[t] / (ms[t] * 1e-3) / 1e12) if ms[t] > 0 else None}
                                     for t in GEMM_TAGS if cnt[t] > 0},
                        # the other side of the step: kernels bound by HBM, as achieved GB/s of their algorithmic bytes
                        'hbm_bound_kernels': {HBM_TAGS[t]: {'bound': 'hbm', 'achieved': by[t] / (ms[t] * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                                                            'unit': 'GB/s', 'frac': by[t] / (ms[t] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                            'traffic': traffic_tab.get(HBM_TAGS[t]), 'launches': cnt[t],
                                                            'avg_launch_us': ms[t] * 1e3 / cnt[t],
                                                            'algorithmic_bytes_per_launch': by[t] / cnt[t]}
                                              for t in HBM_TAGS if cnt[t] > 0 and ms[t] > 0}}
            if account is not None:
                roofline['exclusive_ms_per_step'] = account['per_step_ms']
                roofline['exclusive_covered_ms_per_step'] = account['covered_ms_per_step']
                roofline['exclusive_note'] = account['what']
                roofline['exclusive_truncated_records'] = account['truncated_records']
    # ---- the OTHER arithmetic, same run, same buffers (untimed by the headline): `exact_f32` beside a split-precision headline and vice versa
    other = None
    if pstep is not None and graph is None and not args.no_input_grad and not args.no_other:
        other_mode = 'f32' if precision == 'bf16x3' else 'bf16x3'
        _lib.call('recnow_set_gemm_precision', 1 if other_mode == 'bf16x3' else 0)
        for _ in range(max(rewarm, 5)):
            run_step()
        sync()
        c0 = time.perf_counter()
        for _ in range(args.steps):
            run_step()
        sync()
        tt = torch.tensor([time.perf_counter() - c0], dtype=torch.float64, device=dev)
        if use_dist:
            if args.backend != 'nccl':
                tt = tt.cpu()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        o_el = float(tt.item())
        other = {'gemm_precision': other_mode, 'ms_per_step': o_el * 1e3 / args.steps, 'value': total_rows * args.steps / o_el, 'unit': 'samples/s',
                 'steps': args.steps, 'route': {1: 'row-block persistent kernels', 2: 'row-block persistent forward (k_mix_tile_fwd_s3) + one launch per product in the backward'}.get(pstep.route_code(), 'one launch per product'),
                 'note': 'the same step on the same buffers with the other arithmetic of the products, timed after the headline (its own warm-up, '
                         'barrier + synchronize on both sides, max over the ranks); parity of this mode: tests/test_step_gpu.py, tests/test_northstar_gpu.py'}
        _lib.call('recnow_set_gemm_precision', 1 if precision == 'bf16x3' else 0)
        for _ in range(2):
            run_step()
        sync()
        mark('other precision mode done')
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rank_ms = [elapsed * 1e3 / args.steps]
    if use_dist:
        per_rank = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        per_rank[dist.get_rank()] = elapsed * 1e3 / args.steps
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) for v in per_rank.tolist()]
        if args.backend != 'nccl':
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- parity step (untimed): same shapes and kernels, inputs scaled so that the scores are of O(1) ----------------------
    parity = None
    cpu = None
    if not use_dist and (graph is None or pstep is not None) and not args.no_input_grad:      # under a process group: the cross-rank gate below
        xq = (xd.detach() * CHECK_SCALE).requires_grad_(True)
        loss_q = step(xq)
        torch.cuda.synchronize()
        named = {'cross.' + k: v for k, v in model.cross.named_weights().items()}
        named['head.kernel'], named['head.bias'] = model.head.kernel, model.head.bias
        named_np = {k: v.detach().cpu().numpy() for k, v in named.items()}
        xq_np = x * np.float32(CHECK_SCALE)
        sc_gpu, dx_gpu = last['scores'].detach().cpu().numpy(), (last['dx'] if pstep is not None else xq.grad).cpu().numpy()
        if use_dist:          # the one-collective form runs the backward pass on the unnormalised loss sum: weight gradients are scaled
            dx_gpu = dx_gpu / (np.float32(last['n_pair'].item()) + np.float32(1e-10))      # in the reducer, dx (not a parameter) here
        parity = {'inputs': 'x * %g' % CHECK_SCALE, 'loss': float(loss_q.item()), 'tolerance': PARITY_TOL}
        if True:
            parity['subset_fp64'] = subset_parity(xq_np, groups, labels, named_np, sc_gpu, dx_gpu, int(last['n_pair'].item()))
            worst = max(parity['subset_fp64']['scores'], parity['subset_fp64']['dx'])
            if not args.no_cpu_baseline:
                def compare(c_loss, c_dx, c_grads, c_scores, c_pairs):
                    full = {'loss': rel_err(parity['loss'], c_loss), 'scores': rel_err(sc_gpu, c_scores), 'dx': rel_err(dx_gpu, c_dx),
                            'pairs_equal': int(last['n_pair'].item()) == int(c_pairs)}
                    elem = {fl: {'scores': rel_err_elem(sc_gpu, c_scores, fl), 'dx': rel_err_elem(dx_gpu, c_dx, fl)} for fl, _ in ELEM_GATES}
                    for k, v in named.items():
                        # d loss / d head.bias = sum_i dscore_i cancels to zero: measured on the scale of the other head gradient
                        full[k] = rel_err(v.grad.cpu().numpy(), c_grads[k], scale=float(np.abs(c_grads['head.kernel']).max()) if k == 'head.bias' else None)
                        if k != 'head.bias':
                            for fl, _ in ELEM_GATES:
                                elem[fl][k] = rel_err_elem(v.grad.cpu().numpy(), c_grads[k], fl)
                    # element by element (ELEM_GATES): every entry of at least `floor` x its tensor's largest magnitude, relative to itself
                    full['elementwise'] = {'floor %g (tolerance %g)' % (fl, tol): dict(elem[fl], max=max(elem[fl].values())) for fl, tol in ELEM_GATES}
                    full['elementwise_ok'] = all(max(elem[fl].values()) <= tol for fl, tol in ELEM_GATES)
                    return full
                # (b) the gate: EVERY output and gradient of the full-size step against the fp64 oracle of the whole batch
                o = cpu_step_full(xq_np, groups, labels, named_np, dtype=torch.float64)
                parity['oracle_fp64_full'] = compare(*o[:5])
                worst = max([worst] + [v for k, v in parity['oracle_fp64_full'].items() if k not in ('pairs_equal', 'elementwise', 'elementwise_ok')])
                if not parity['oracle_fp64_full']['pairs_equal'] or not parity['oracle_fp64_full']['elementwise_ok']:
                    worst = float('inf')
                del o
                # (c) the CPU baseline: the same step by the fp32 port, timed (reported beside: fp32 against fp32, not part of the gate)
                c_loss, c_dx, c_grads, c_scores, c_pairs, c_sec = cpu_step_full(xq_np, groups, labels, named_np, warm_rows=4096)
                parity['cpu_port_fp32'] = compare(c_loss, c_dx, c_grads, c_scores, c_pairs)
                cpu = {'value': rows / c_sec, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                       'sample': '1 step of fwd+bwd at the full batch B=%d (same inputs, weights and 64 rows/group as the parity step), after an '
                                 'untimed warm-up over 4096 rows: oracle/dense_ref.py layers in %d-row chunks on torch-CPU fp32 (scores pass, '
                                 'then forward+backward per chunk) + segment-based C pair loss with OpenMP (oracle/pairs_oracle.c), %.1f s; '
                                 'TF2 itself is not installable here' % (rows, 8192, c_sec),
                       'dense_b8192': cpu_dense_b8192()}
            parity['parity_max_rel'] = worst
            parity['ok'] = bool(worst <= PARITY_TOL and abs(parity['loss'] - float(np.log(2.0))) > 1e-3)

    # ---- N > 1 (or --force-dist): the same gate, across the ranks -------------------------------------------------------------
    # One more step on the scaled inputs; (scores, labels, groups) of every rank are all-gathered ONCE and the reduced loss and the
    # pair count are held to oracle/pairs_oracle.c on the gathered batch; every rank evaluates the fp64 oracle of the layers on ITS
    # rows (scores -> gathered -> the global pair gradient -> backward of its rows), the per-rank weight gradients are summed
    # (all-reduce of the fp64 oracle gradients) and the all-reduced GPU gradients are held to that sum.
    if use_dist and (graph is None or pstep is not None) and not args.no_input_grad:
        _, PO = _oracle()
        rccl_ranks = dist.get_world_size()
        torch.set_num_threads(max(1, min(os.cpu_count() or 1, 2 * cpu_share()) // max(world, 1)))
        xq = (xd.detach() * CHECK_SCALE).requires_grad_(True)
        loss_q = step(xq)
        torch.cuda.synchronize()
        p_glob = float(last['n_pair'].item())
        named = {'cross.' + k: v for k, v in model.cross.named_weights().items()}
        named['head.kernel'], named['head.bias'] = model.head.kernel, model.head.bias
        named_np = {k: v.detach().cpu().numpy() for k, v in named.items()}
        xq_np = x * np.float32(CHECK_SCALE)
        sc_gpu = last['scores'].detach().cpu().numpy()
        dx_gpu = (last['dx'] if pstep is not None else xq.grad).cpu().numpy() / (np.float32(p_glob) + np.float32(1e-10))       # every N > 1 route differentiates the loss SUM
        full_gate = not args.no_cpu_baseline
        fwd, wleaf = _cpu_model(named_np, torch.float64)
        sc64 = cpu_scores(fwd, xq_np, torch.float64) if full_gate else sc_gpu.astype(np.float64)

        def gather(a):
            a = np.ascontiguousarray(a)
            t = torch.zeros(max(rank_rows), dtype=torch.from_numpy(a[:0]).dtype)      # shards may be ragged (--shard hash): padded to the longest
            t[:a.shape[0]] = torch.from_numpy(a)
            if args.backend == 'nccl':          # (gloo gathers host tensors)
                t = t.to(dev)
            parts = [torch.empty_like(t) for _ in range(rccl_ranks)]
            dist.all_gather(parts, t)
            return torch.cat([parts[r][:rank_rows[r]] for r in range(rccl_ranks)]).cpu().numpy()
        g_all, y_all, s_all, s64_all = gather(groups), gather(labels), gather(sc_gpu), gather(sc64)
        # (a) the loss stage + the reduction: oracle pair loss of the GATHERED batch on the GPU's own scores
        o_loss, _, o_pairs = PO.pairwise_bpr(g_all, y_all, s_all, grouped=True)
        parity = {'inputs': 'x * %g' % CHECK_SCALE, 'loss': float(loss_q.item()), 'tolerance': PARITY_TOL, 'rccl_ranks': rccl_ranks,
                  'gathered_rows': int(g_all.size),
                  'gathered_batch_pairs_oracle': {'loss': rel_err(float(loss_q.item()), o_loss), 'pairs_gpu': int(p_glob), 'pairs_oracle': int(o_pairs),
                                                  'pairs_equal': int(p_glob) == int(o_pairs)}}
        worst = parity['gathered_batch_pairs_oracle']['loss']
        ok = int(p_glob) == int(o_pairs)
        # ~256 rows of whole groups of THIS rank's shard against the fp64 oracle (scores, d loss / d x), global pair count
        sub = subset_parity(xq_np, groups, labels, named_np, sc_gpu, dx_gpu, int(p_glob), p_total=int(o_pairs))
        loc = torch.tensor([sub['scores'], sub['dx']], dtype=torch.float64, device=dev)
        dist.all_reduce(loc, op=dist.ReduceOp.MAX)
        parity['subset_fp64'] = dict(sub, scores=float(loc[0].item()), dx=float(loc[1].item()), note='max over the ranks')
        worst = max(worst, parity['subset_fp64']['scores'], parity['subset_fp64']['dx'])
        if full_gate:
            # (b) the whole chain in fp64: global pair gradient from the gathered fp64 scores, backward of the local rows, gradients summed
            f_loss, ds_all, f_pairs = PO.pairwise_bpr(g_all, y_all, s64_all, grouped=True)
            lo = sum(rank_rows[:rank])
            dx64 = cpu_backward(fwd, xq_np, ds_all[lo:lo + rows], torch.float64)
            names = sorted(named)
            flat = torch.cat([torch.from_numpy(wleaf[k].grad.numpy().reshape(-1)) for k in names]).to(dev)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat = flat.cpu().numpy()
            loc = torch.tensor([rel_err(sc_gpu, sc64, scale=np.abs(s64_all).max()), rel_err(dx_gpu, dx64)], dtype=torch.float64, device=dev)
            dist.all_reduce(loc, op=dist.ReduceOp.MAX)
            full = {'loss': rel_err(float(loss_q.item()), f_loss), 'pairs_equal': int(p_glob) == int(f_pairs),
                    'scores': float(loc[0].item()), 'dx': float(loc[1].item())}
            off = 0
            grads64 = {}
            for k in names:
                n = named[k].numel()
                grads64[k] = flat[off:off + n].reshape(tuple(named[k].shape))
                off += n
            hk_scale = float(np.abs(grads64['head.kernel']).max())
            for k in names:
                full[k] = rel_err(named[k].grad.cpu().numpy(), grads64[k], scale=hk_scale if k == 'head.bias' else None)
            parity['oracle_fp64_all_ranks'] = full
            worst = max([worst] + [v for k, v in full.items() if k != 'pairs_equal'])
            ok = ok and full['pairs_equal']
        parity['parity_max_rel'] = worst
        parity['ok'] = bool(ok and worst <= PARITY_TOL and abs(parity['loss'] - float(np.log(2.0))) > 1e-3)

    if rank == 0:
        out = {
            'metric': 'samples/sec fwd+bwd, in-batch pairwise + DCN-v2, B=65536 at 1/2/4/8 GPUs',
            'value': total_rows * args.steps / elapsed,
            'unit': 'samples/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps,
            'higher_is_better': True,
            'scaling': args.scaling if args.rows is None else 'weak',
            'shard': args.shard,
            'vs_baseline': None,
            'dtype': 'f32' if precision == 'f32' else 'f32 via 3 x bf16 six-term split (bf16 MFMA, f32 accumulate; per-product error <= 2^-25)',
            'data': 'synthetic',
            'config': {'workload': 'configs[2]: dcn_mix_layer (3 cross layers, low-rank 64, 2 experts) + MultiDense(1,1) head + '
                                   'in-batch pairwise (logistic), global B=%d = %s rows on %d GPU(s), 64 fields x 16-dim, ~64 rows/group'
                                   % (total_rows, ('%d' % rows) if len(set(rank_rows)) == 1 else '/'.join(str(r) for r in rank_rows), world),
                       'rows_per_gpu': rows, 'rows_per_rank': rank_rows,
                       'global_batch': total_rows, 'input_grad': not args.no_input_grad, 'hip_graph': bool(use_graph), 'parallelism': 'dp%d' % world,
                       'route': ('whole-step entry recnow_dcn_mix_step (one C call per phase; cross layers: %s; grouping of the batch: %s)'
                                 % ({1: 'row-block persistent kernels k_mix_tile_fwd / k_mix_tile_bwd',
                                     2: 'forward: ONE row-block persistent launch on the bf16 MFMA (k_mix_tile_fwd_s3); backward: one launch per product (k_gemm_s3 / k_gemm_split / k_gemm_shortk)'}
                                    .get(pstep.route_code(), 'one launch per product (k_gemm / k_gemm_shortk)'),
                                    {'side': 'on a side stream under the forward pass', 'inline': 'on the main stream in front of the forward pass',
                                     'after': 'on the main stream behind the forward pass'}.get(getattr(pstep, 'group_mode', 'side'), '?'))
                                 + (', replayed from HIP graphs' if use_graph else '') + (', weight-gradient products on a second stream' if two_streams else '')
                                 + (', ragged batch on padded storage (%d -> %d rows)' % (pstep.B, pstep.B_pad) if pstep.B_pad != pstep.B else '')) if use_step else
                                'fused node dcn_mix_score through autograd + grouping on a side stream' if fused else ('drop-in layers' if args.unfused else 'drop-in layers (fused route not available)'),
                       'gemm_precision': precision, 'gemm_precision_rule': args.gemm_precision if args.gemm_precision != 'auto' else
                       'auto: bf16x3 from %d rows per GPU, f32 below (the faster one at each shard size, both under the 1e-5 gate)' % SPLIT_FROM_ROWS,
                       'loss': float(loss.item()), 'host_enqueue_ms_per_step': host_ms, 'untimed_steps': args.warmup + rewarm,
                       'grads_copied_into_buckets': getattr(layerwise, 'last_foreign', None) if layerwise is not None else None},
            'roofline': roofline,
            'parity': parity,
            'parity_max_rel': parity.get('parity_max_rel') if parity else None,
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
        if other is not None:           # (last key of the line: it survives in a tail of the output)
            out['exact_f32' if other['gemm_precision'] == 'f32' else 'split_precision'] = other
        if use_dist:
            out['rccl_ranks'] = parity['rccl_ranks'] if parity and 'rccl_ranks' in parity else dist.get_world_size()
            out['backend'] = args.backend
            out['ms_per_step_per_rank'] = {'min': min(rank_ms), 'max': max(rank_ms), 'ranks': rank_ms}
            if comm is not None:
                out['comm_exposed_ms'] = comm['comm_exposed_ms']
                out['comm'] = comm
        if other is not None:           # keep it the LAST key of the line
            k = 'exact_f32' if other['gemm_precision'] == 'f32' else 'split_precision'
            out[k] = out.pop(k)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (C stdio is flushed first: anything still buffered belongs to stderr's side of the redirection above)
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())


if __name__ == '__main__':
    main()
