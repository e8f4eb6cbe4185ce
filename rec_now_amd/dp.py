"""Batch data-parallelism for the hot path: one process per GPU, torch.distributed ('nccl' = RCCL over xGMI on ROCm;
'gloo' in the CPU tests).  The reference has no distributed code at all; this is the only strategy
the path needs:

  * the layers are row-independent (weights replicated)            -> shard rows, SUM-all-reduce weight gradients;
  * the losses couple only rows of the same group                  -> every group must live on ONE rank
    (`shard_rows_by_group`), then the global loss is  all_reduce(sum_p w_p l_p) / (all_reduce(P) + 1e-10)
    and each rank's local gradient only needs the global pair count (one 2-float all-reduce, no data-path exchange).

With that sharding the N-GPU loss and gradients equal the 1-GPU ones at the global batch.
"""
import torch
import torch.distributed as dist

SMALL_POSIVITE_FLOAT = 1.0e-10


import os as _os
_SKIP_COLLECTIVE = _os.environ.get('RECNOW_DP_SKIP_COLLECTIVE') == '1'      # diagnostics only: price the path around the collective
FORCE_COLLECTIVES = False      # diagnostics: run the collectives even in a 1-rank process group (bench.py --force-dist)


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def shard_rows_by_group(group_ids, world_size):
    """Owner rank of every row: all rows of a group go to one rank (hash of the id).  group_ids: integer tensor/array.
    Returns an int64 tensor of ranks; rank r keeps rows where result == r."""
    g = torch.as_tensor(group_ids).to(torch.int64)
    h = (g * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF     # Fibonacci hashing, wraps in int64
    return (h >> 17) % world_size


def global_pairwise_loss(local_loss_sum, local_n_pair):
    """Combine per-rank pair-loss SUMS into the global mean loss.

    local_loss_sum: 0-dim tensor = sum_p w_p*l_p over THIS rank's pairs (e.g. `pairwise_loss_fused(..., reduce_mean=
    False)`), attached to the autograd graph; local_n_pair: 0-dim float tensor.
    Returns (loss_for_backward, global_loss_value, global_n_pair):
      loss_for_backward = local_loss_sum / (P_global + 1e-10)   -- backward gives exactly this rank's share of dL/ds
      global_loss_value = all_reduce(local sums) / (P_global + 1e-10) (detached; identical on every rank)."""
    stats = torch.stack([local_loss_sum.detach().to(torch.float32), local_n_pair.detach().to(torch.float32)])
    if is_dist():
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    denom = stats[1] + SMALL_POSIVITE_FLOAT
    return local_loss_sum / denom, stats[0] / denom, stats[1]


def global_listwise_loss(local_loss_sum, local_n_valid):
    """Same for the listwise loss: mean over ALL valid lists of all ranks; 0 when there is none (nan_to_zero)."""
    stats = torch.stack([local_loss_sum.detach().to(torch.float32), local_n_valid.detach().to(torch.float32)])
    if is_dist():
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    denom = torch.clamp(stats[1], min=1.0)
    return local_loss_sum / denom, stats[0] / denom, stats[1]


def gathered_pairwise_loss(outputs, labels, groups, loss_fn=None, **kwargs):
    """The EXACT in-batch pairwise loss for rows that arrive ARBITRARILY sharded (groups split across ranks), SURVEY.md section 8e
    variant C3: one all-gather of the (score, label, group) triples (12 bytes per row: 0.8 MB at B = 65 536), then every rank
    evaluates the loss of the GLOBAL batch on the gathered data -- the pair stage is ~0.1 ms at 65 536 rows, cheaper than any
    exchange of partial results -- and keeps the gradient of ITS OWN rows: no return exchange.  The pairs are exactly those of
    /root/reference/rec_now/rec_block/pairwise_loss_from_batch.py:254-274 on the concatenated batch.

    outputs (B_local,) with grad; labels, groups (B_local,).  loss_fn(outputs, labels, groups, **kwargs) -> scalar loss: default
    `rec_block.pairwise_loss_from_batch.pairwise_loss` (GPU kernels); the CPU tests pass the oracle's.  Shards may be uneven or empty.
    Returns the global mean loss (identical on every rank); its backward gives d loss_global / d outputs for the local rows, so
    weight gradients only need the usual SUM all-reduce (`GradientAllReducer.all_reduce()`, no 1/P scaling afterwards).
    Without a process group this is `loss_fn(outputs, labels, groups)`."""
    if loss_fn is None:
        from .rec_block.pairwise_loss_from_batch import pairwise_loss as loss_fn
    flat = outputs.reshape(-1)
    # the reference accepts ONE group tensor or a LIST of them (pairs must agree on every one: pairwise_loss_from_batch.py:65-73)
    as_list = isinstance(groups, (list, tuple))
    group_list = [torch.as_tensor(g).reshape(-1) for g in (groups if as_list else [groups])]
    if not group_list:
        raise ValueError('groups: an empty list')
    n_local = flat.numel()
    for g in group_list:
        if g.numel() != n_local:
            raise ValueError('groups: %d ids for %d outputs' % (g.numel(), n_local))
    if labels.numel() != n_local:
        raise ValueError('labels: %d values for %d outputs' % (labels.numel(), n_local))
    regroup = (lambda gs: list(gs)) if as_list else (lambda gs: gs[0])      # noqa: E731
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)):
        return loss_fn(flat, labels.reshape(-1), regroup(group_list), **kwargs)
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = torch.zeros(world, dtype=torch.int64, device=flat.device)
    counts[rank] = n_local
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    counts = [int(c) for c in counts.tolist()]
    n_max = max(max(counts), 1)

    def gather_rows(vs):
        """ONE all-gather of several ragged (n_local,) vectors of one dtype, stacked as the rows of a (len(vs), n_max) block (zero padded to the
        longest shard): returns the gathered vectors in the order given."""
        block = torch.zeros((len(vs), n_max), dtype=vs[0].dtype, device=flat.device)
        for i, v in enumerate(vs):
            block[i, :n_local] = v.reshape(-1).to(flat.device)
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block)
        return [torch.cat([parts[r][i, :counts[r]] for r in range(world)]) for i in range(len(vs))]
    # Every tensor travels in its NATIVE dtype -- a float block would merge distinct int32 ids above 2^24 (float32) and hashed int64 ids above
    # 2^53 (float64) after the gather: silently wrong pairs -- but tensors of EQUAL dtype share one collective (round 5; round 4 issued one
    # blocking all-gather per tensor: 2 + len(groups) of them on the small shards where latency is the cost): the usual case -- float32
    # scores, labels and group ids -- is ONE all-gather behind the counts all-reduce, integer ids make it two.
    items = [('s', flat.detach()), ('y', labels.reshape(-1))] + [('g%d' % i, g) for i, g in enumerate(group_list)]
    by_dtype = {}
    for name, v in items:
        by_dtype.setdefault(v.dtype, []).append((name, v))
    gathered = {}
    for dt in by_dtype:
        names = [n for n, _ in by_dtype[dt]]
        for n, t in zip(names, gather_rows([v for _, v in by_dtype[dt]])):
            gathered[n] = t
    s_all = gathered['s'].requires_grad_(True)
    y_all = gathered['y']
    g_all = regroup([gathered['g%d' % i] for i in range(len(group_list))])
    with torch.enable_grad():
        loss_all = loss_fn(s_all, y_all, g_all, **kwargs)
        (ds_all,) = torch.autograd.grad(loss_all, s_all, allow_unused=True)
    lo = sum(counts[:rank])
    ds_mine = (torch.zeros(n_local, dtype=flat.dtype, device=flat.device) if ds_all is None else ds_all[lo:lo + n_local]).detach()
    # value = the global loss; gradient w.r.t. the local outputs = this rank's rows of d loss_global / d scores
    return loss_all.detach() + ((flat - flat.detach()) * ds_mine).sum()


class GradientAllReducer(object):
    """SUM-all-reduce of the weight gradients in a few large flat buckets (xGMI is point-to-point: 7 links x ~153 GB/s
    per GPU, ring collectives are per-link bound, so fewer/larger messages win; the hot path's gradients are 3-80 MB)."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._extra = None
        self._extra_out = None

    def all_reduce_with_loss(self, local_loss_sum, local_count, eps=SMALL_POSIVITE_FLOAT):
        """ONE collective per step.  The pairwise / listwise loss is linear in 1/P_global, so instead of all-reducing
        (loss sum, P) before the backward pass (a sync point between forward and backward, `global_pairwise_loss`), run
        the backward pass on the UNNORMALISED local loss sum, then call this: the two statistics ride in the last gradient
        bucket, and every gradient is divided by (P_global + eps) afterwards (on the flat buffers).
        Returns (global mean loss, P_global) as 0-dim tensors (detached).  Works without a process group too."""
        stats = torch.stack([local_loss_sum.detach().to(torch.float32), local_count.detach().to(torch.float32)])
        if is_dist():
            self._extra = stats
            self.all_reduce(scale=lambda ex: 1.0 / (ex[1] + eps))
            stats = self._extra_out
            self._extra = None
        else:
            inv = 1.0 / (stats[1] + eps)
            grads = [p.grad for p in self.params if p.grad is not None]
            if grads:
                torch._foreach_mul_(grads, inv)
        return stats[0] / (stats[1] + eps), stats[1]

    def all_reduce(self, async_op=False, scale=None):
        """Call after backward.  Gradients missing on this rank count as zero.  Packing / unpacking of a bucket is one
        `torch.cat` and one multi-tensor copy (not a kernel per parameter).  `scale`: optional callable(extra_out) -> 0-dim
        tensor the reduced gradients are multiplied with (on the flat buffer, one kernel)."""
        if not is_dist():
            return []
        works = []
        extra = self._extra
        for i, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket]
            if extra is not None and i == len(self.buckets) - 1:
                parts.append(extra.to(parts[0].dtype).reshape(-1))
            total = sum(t.numel() for t in parts)
            flat = self._flat[i]
            if flat is None or flat.numel() != total or flat.device != bucket[0].device:
                flat = torch.empty(total, dtype=bucket[0].dtype, device=bucket[0].device)
                self._flat[i] = flat
            torch.cat(parts, out=flat)
            works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), i))
        if async_op:
            return works
        self.finish(works, scale)
        return []

    def finish(self, works, scale=None):
        for work, i in works:
            work.wait()
        for work, i in reversed(works):              # the last bucket carries the extras the scale may depend on
            flat = self._flat[i]
            n = sum(p.numel() for p in self.buckets[i])
            if flat.numel() > n:
                self._extra_out = flat[n:].clone()
        factor = scale(self._extra_out) if scale is not None else None
        for work, i in works:
            flat, off = self._flat[i], 0
            if factor is not None:
                flat.mul_(factor)
            dst, src = [], []
            for p in self.buckets[i]:
                k = p.numel()
                view = flat[off:off + k].reshape(p.shape)
                if p.grad is None:
                    p.grad = view.clone()
                else:
                    dst.append(p.grad)
                    src.append(view)
                off += k
            if dst:
                torch._foreach_copy_(dst, src)


class LayerwiseReducer(object):
    """Gradient all-reduce overlapped with the backward pass, one bucket per cross layer.

    `stages`: parameter lists in the order their gradients become final in the backward pass (for the north-star model: the
    top cross layer + the scoring head first, layer 0 last).  `events[i]` (fused.GpuEvent) is recorded by
    `recnow_dcn_mix_score_bwd` once every gradient of stage i has been issued.  Every stage owns ONE flat fp32 bucket;
    `buffer_of(param)` is that parameter's slice of it, to be handed to the backward pass as gradient storage
    (`fused.dcn_mix_score(..., grad_buffers=...)`), so the collective runs in place: nothing is packed before and nothing is
    copied after it (a gradient that was produced elsewhere is copied in and out, as a fallback).
    After `loss_sum.backward()` has returned (all kernels are enqueued, none need have run), `reduce(loss_sum, count)` makes a
    side stream wait for event i, all-reduce bucket i and scale it -- while the main stream is still executing the backward of
    the stages below.  The two loss statistics ride in the FIRST bucket (they are known before the backward pass starts), so
    every later bucket is multiplied by 1 / (P_global + eps) as soon as its own collective has finished.  Works without a
    process group (scaling only).  Returns (global mean loss, P_global) as 0-dim tensors.

    Gradient contract: after a step `p.grad` IS a view of the bucket.  The usual loop (`p.grad = None` / `zero_grad(set_to_none=
    True)` before the next backward) lets the next backward write the bucket in place.  When a gradient is still alive at the next
    forward (gradient accumulation), `fused.dcn_mix_score` gives that parameter a fresh gradient tensor instead, autograd adds it
    to the live `p.grad` (= the bucket view), and `reduce` all-reduces the accumulated sum -- note that `reduce` scales the whole
    bucket by 1 / (P_global + eps) each time it runs, so accumulate with ONE `reduce` at the end of the accumulation window."""

    def __init__(self, stages, events, device, one_collective=None):
        self.stages = [[p for p in stage if p.requires_grad] for stage in stages]
        self.events = list(events)
        if len(self.events) != len(self.stages):
            raise ValueError('one event per stage')
        dev = torch.device(device)
        self.comm = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None      # CPU (gloo tests): in order
        # one_collective (default: RECNOW_DP_ONE_BUCKET=1 in the environment, else off): ONE all-reduce over all stages' gradients behind the last
        # stage instead of one per stage under the backward pass.  Per-stage collectives hide all but the last bucket's -- when something can run
        # beside the backward launches.  The row-block backward chain of the shard sizes is ONE launch that holds every CU, followed by ~140 us of
        # weight-gradient products: the stages' events then fire within ~40 us of each other at the end of the pass, so
        # three 1 MB ring all-reduces (latency-bound: ~3 x (2 (N-1) hops x ~2 us + 2 x 1.06 MB (N-1)/N / 153 GB/s) each) queue up behind the
        # step, where one 3.2 MB all-reduce pays the hop latency once.  Which wins is a property of the node: the switch makes it an A/B.
        # Applies to the IN-PLACE protocol only (`stage_done` / `reduce_in_place`, what step.DCNMixPairwiseStep drives): the autograd-route `reduce()` keeps
        # one collective per stage whatever this says.
        self.one_collective = (_os.environ.get('RECNOW_DP_ONE_BUCKET') == '1') if one_collective is None else bool(one_collective)
        self._flat, self._view = [], {}
        # the stages' buckets are slices of ONE allocation (each starting on a 16-byte boundary): per-stage collectives see their own
        # slice, the one-collective form the whole of it
        sizes = [sum(p.numel() for p in stage) + (2 if i == 0 else 0) for i, stage in enumerate(self.stages)]
        starts, total = [], 0
        for n in sizes:
            starts.append(total)
            total += -(-n // 4) * 4
        self._all = torch.zeros(max(total, 4), dtype=torch.float32, device=dev)
        for i, stage in enumerate(self.stages):
            flat = self._all[starts[i]:starts[i] + sizes[i]]
            off = 0
            for p in stage:
                self._view[id(p)] = flat[off:off + p.numel()]
                off += p.numel()
            self._flat.append(flat)
        self._rest = self._all[-(-sizes[0] // 4) * 4:] if len(sizes) > 1 else self._all[:0]      # everything behind the first stage (+ its statistics)

    def buffer_of(self, param):
        """The slice of its stage's flat bucket that holds `param`'s gradient (1-D view), or None for a foreign parameter."""
        return self._view.get(id(param))

    # ---- in-place protocol of step.DCNMixPairwiseStep: gradients AND statistics are produced inside the buckets ---------------
    def stats_slot(self):
        """The two floats behind the first bucket's gradients: {local loss sum, local pair count} are written there by the step's
        loss stage (`recnow_dcn_mix_step_desc.stats`), so they ride in the first collective without a packing kernel."""
        return self._flat[0][-2:]

    def stage_done(self, i, eps=SMALL_POSIVITE_FLOAT, recorded=False):
        """Every gradient of stage i (and, for i == 0, the statistics) has been ENQUEUED on the current stream: mark that point and
        enqueue the stage's all-reduce + 1 / (P_global + eps) scaling on the communication stream.  The current stream is not made to
        wait: the backward of the stages below keeps running under the collective.
        recorded=True: the library has already recorded the stage's event where the stage's last gradient was issued (the
        `layer_events_host` of recnow_dcn_mix_step / recnow_dcn_mix_score_bwd): only the wait and the collective are enqueued."""
        flat = self._flat[i]
        n_grad = flat.numel() - (2 if i == 0 else 0)
        stats = self._flat[0][-2:]
        last = i == len(self._flat) - 1
        if self.comm is None:                       # CPU tensors (gloo tests): in order on the host
            if self.one_collective:
                if not last:
                    return
                if is_dist() and not _SKIP_COLLECTIVE:
                    dist.all_reduce(self._all, op=dist.ReduceOp.SUM)
                inv = 1.0 / (stats[1] + eps)
                self._result = torch.stack([stats[0] * inv, stats[1]])
                n0 = self._flat[0].numel() - 2
                self._flat[0][:n0].mul_(inv)
                self._rest.mul_(inv)
                return
            if is_dist() and not _SKIP_COLLECTIVE:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat[:n_grad].mul_(1.0 / (stats[1] + eps))
            if i == 0:
                self._result = torch.stack([stats[0] / (stats[1] + eps), stats[1]])
            return
        from . import _lib
        main = torch.cuda.current_stream()
        ev = self.events[i]
        if not recorded:
            _lib.call('recnow_event_record', ev.handle, _lib._P(main.cuda_stream))
        if getattr(self, '_result', None) is None or not self._result.is_cuda:
            self._result = torch.empty(2, dtype=torch.float32, device=flat.device)
        if self.one_collective:
            if not last:
                return
            for e in self.events:                   # the stages' events are recorded on different streams: wait for every one of them
                e.wait(self.comm)
            if is_dist() and not _SKIP_COLLECTIVE:
                torch.cuda.set_stream(self.comm)
                try:
                    dist.all_reduce(self._all, op=dist.ReduceOp.SUM)
                finally:
                    torch.cuda.set_stream(main)
            n0 = self._flat[0].numel() - 2
            _lib.call('recnow_scale_by_inv_count', _lib.ptr(self._flat[0]), n0, _lib.ptr(stats[1:]), float(eps), _lib.ptr(stats), _lib.ptr(self._result),
                      _lib._P(self.comm.cuda_stream))
            if self._rest.numel():
                _lib.call('recnow_scale_by_inv_count', _lib.ptr(self._rest), self._rest.numel(), _lib.ptr(stats[1:]), float(eps), None, None,
                          _lib._P(self.comm.cuda_stream))
            return
        ev.wait(self.comm)
        if is_dist() and not _SKIP_COLLECTIVE:
            torch.cuda.set_stream(self.comm)        # (the context manager costs ~40 us of host time per use)
            try:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            finally:
                torch.cuda.set_stream(main)
        _lib.call('recnow_scale_by_inv_count', _lib.ptr(flat), n_grad, _lib.ptr(stats[1:]), float(eps), _lib.ptr(stats) if i == 0 else None,
                  _lib.ptr(self._result) if i == 0 else None, _lib._P(self.comm.cuda_stream))

    def reduce_in_place(self):
        """After the last `stage_done`: the current stream waits for the communication stream (the gradients are final for whatever
        comes next).  Returns (global mean loss, P_global) as 0-dim device tensors.  They are VIEWS of one persistent 2-float buffer that
        the next step's first `stage_done` overwrites on the communication stream (unlike `reduce()`, which allocates its result):
        `.clone()` them to keep a value across steps."""
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        return self._result[0], self._result[1]

    def prepare(self, local_loss_sum, local_count):
        """Call BEFORE `local_loss_sum.backward()`: packs the two loss statistics on the current stream and marks that point, so
        that `reduce` has nothing to wait for on the main stream but the per-stage events (waiting for the main stream itself
        would wait for the whole backward pass, i.e. no overlap)."""
        # straight into the tail of the first bucket (one small kernel on the main stream, before the backward pass)
        self._stats = torch.stack([local_loss_sum.detach().to(torch.float32), local_count.detach().to(torch.float32)], out=self._flat[0][-2:])
        self._stats_ready = None
        if self.comm is not None:
            self._stats_ready = torch.cuda.Event()
            self._stats_ready.record(torch.cuda.current_stream())

    def reduce(self, local_loss_sum, local_count, eps=SMALL_POSIVITE_FLOAT):
        import contextlib
        main = torch.cuda.current_stream() if self.comm is not None else None
        prepared = getattr(self, '_stats', None) is not None
        stats = self._stats if prepared else torch.stack([local_loss_sum.detach().to(torch.float32), local_count.detach().to(torch.float32)])
        stats_ready = self._stats_ready if prepared else None
        self._stats = None
        if not is_dist():
            inv = 1.0 / (stats[1] + eps)
            grads = [p.grad for stage in self.stages for p in stage if p.grad is not None]
            if grads:
                torch._foreach_mul_(grads, inv)
            return stats[0] * inv, stats[1]
        if self.comm is not None:
            if stats_ready is not None and all(e is not None for e in self.events):
                self.comm.wait_event(stats_ready)       # only the statistics; the gradients are ordered by the per-stage events
            else:
                self.comm.wait_stream(main)             # no events / no prepare(): everything enqueued so far, i.e. after the backward pass
        out_stats = None
        self.last_foreign = 0                           # diagnostics: gradients that had to be copied into their bucket
        with (torch.cuda.stream(self.comm) if self.comm is not None else contextlib.nullcontext()):
            for i, stage in enumerate(self.stages):
                if self.events[i] is not None and self.comm is not None:
                    self.events[i].wait(self.comm)
                flat = self._flat[i]
                foreign = []                            # gradients that were not written into the bucket by the backward pass
                for p in stage:
                    view = self._view[id(p)]
                    if p.grad is None:
                        view.zero_()
                        p.grad = view.view(p.shape)
                    elif p.grad.data_ptr() != view.data_ptr():
                        view.copy_(p.grad.reshape(-1))
                        foreign.append((p, view))
                        self.last_foreign += 1
                if i == 0 and not prepared:
                    flat[-2:].copy_(stats)
                if not _SKIP_COLLECTIVE:
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                n_grad = flat.numel() - (2 if i == 0 else 0)        # the statistics themselves stay unscaled in the first bucket's tail
                if i == 0:
                    out_stats = flat[-2:]
                if flat.is_cuda:            # one small launch (torch's broadcast multiply by a 0-dim tensor takes ~30 us per bucket);
                    from . import _lib      # the first bucket's launch also leaves (mean loss, P) = what reduce() returns
                    if i == 0:
                        result = torch.empty(2, dtype=torch.float32, device=flat.device)
                    _lib.call('recnow_scale_by_inv_count', _lib.ptr(flat), n_grad, _lib.ptr(out_stats[1:]), float(eps),
                              _lib.ptr(out_stats) if i == 0 else None, _lib.ptr(result) if i == 0 else None,
                              _lib._P(torch.cuda.current_stream().cuda_stream))
                else:
                    flat[:n_grad].mul_(1.0 / (out_stats[1] + eps))
                    if i == 0:
                        result = torch.stack([out_stats[0] / (out_stats[1] + eps), out_stats[1]])
                for p, view in foreign:
                    if self.comm is not None:
                        p.grad.record_stream(self.comm)
                    p.grad.copy_(view.view(p.shape))
        if self.comm is not None:
            main.wait_stream(self.comm)                 # the gradients are final for whatever the main stream does next
            result.record_stream(main)
        return result[0], result[1]


class OverlappedGradientReducer(object):
    """Gradient all-reduce UNDER an autograd backward pass, for models whose backward is not one library call (round 6: the configs[3] / configs[4]
    model steps -- CIN || FM -> head -> pairwise, PLE -> heads -> listwise).  `GradientAllReducer` packs and reduces everything AFTER the backward
    pass; here the parameters are bucketed in REVERSE registration order (the order their gradients become final: the head first, the first
    layer last), every parameter carries a post-accumulate-grad hook that copies its gradient into its bucket's flat buffer, and the hook that
    completes a bucket records an event and enqueues that bucket's all-reduce on the communication stream -- the backward of the layers below
    runs on beside it.  The two loss statistics ride in the FIRST bucket to go (they are known before the backward pass starts: `prepare`).

        reducer.prepare(local_loss_sum, local_count)      # before backward
        local_loss_sum.backward()                         # hooks fire; collectives start while the pass runs
        loss, count = reducer.finish()                    # waits, scales every gradient by 1 / denom(count_global), p.grad = bucket views

    denom: 'eps' -> count + 1e-10 (the pairwise mean, rec_block/pairwise_loss_from_batch.py:279); 'max1' -> max(count, 1) (the listwise mean over
    valid lists with NaN -> 0, rec_block/listwise_loss_from_batch.py:151-173).  Works without a process group (scaling only) and on CPU tensors
    (gloo tests: the collectives run in the hooks, in order).  Set `p.grad = None` (or zero_grad(set_to_none=True)) before each step: a live
    gradient would be accumulated into by autograd and reduced a second time."""

    def __init__(self, params, bucket_bytes=24 << 20, denom='eps'):
        if denom not in ('eps', 'max1'):
            raise ValueError("denom: 'eps' or 'max1'")
        self.denom = denom
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no parameter requires a gradient')
        dev, dt = self.params[0].device, self.params[0].dtype
        order = list(reversed(self.params))
        groups, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        self.buckets = groups
        self._flat, self._slot = [], {}
        for bi, bucket in enumerate(groups):
            total = sum(-(-p.numel() // 4) * 4 for p in bucket) + (4 if bi == 0 else 0)      # 16-byte aligned slices; the statistics behind bucket 0
            flat = torch.zeros(total, dtype=dt, device=dev)
            off = 0
            for p in bucket:
                self._slot[id(p)] = (bi, flat[off:off + p.numel()])
                off += -(-p.numel() // 4) * 4
            self._flat.append(flat)
        self.comm = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self._events = [torch.cuda.Event() for _ in groups] if self.comm is not None else None
        self._arrived = [0] * len(groups)
        self._works = []
        self._armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def prepare(self, local_loss_sum, local_count):
        torch.stack([local_loss_sum.detach().to(self._flat[0].dtype), local_count.detach().to(self._flat[0].dtype)], out=self._flat[0][-4:-2])
        self._arrived = [0] * len(self.buckets)
        self._works = []
        self._armed = True

    def _launch(self, bi):
        flat = self._flat[bi]
        if not is_dist() or _SKIP_COLLECTIVE:
            return
        if self.comm is None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            return
        main = torch.cuda.current_stream()
        self._events[bi].record(main)
        self.comm.wait_event(self._events[bi])
        torch.cuda.set_stream(self.comm)
        try:
            self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        finally:
            torch.cuda.set_stream(main)

    def _on_grad(self, p):
        if not self._armed or p.grad is None:
            return
        bi, view = self._slot[id(p)]
        view.copy_(p.grad.reshape(-1))
        self._arrived[bi] += 1
        if self._arrived[bi] == len(self.buckets[bi]):
            self._launch(bi)

    def finish(self):
        """After backward(): buckets whose parameters got no gradient on this rank are reduced now (zeros), every collective is waited for, the
        gradients are scaled and `p.grad` becomes the parameter's slice of its bucket.  Returns (global mean loss, global count), 0-dim tensors."""
        if not self._armed:
            raise RuntimeError('finish() without prepare()')
        self._armed = False
        for bi, bucket in enumerate(self.buckets):
            if self._arrived[bi] < len(bucket):
                for p in bucket:
                    if p.grad is None:
                        self._slot[id(p)][1].zero_()
                    elif self._arrived[bi] == 0 or p.grad.data_ptr() != self._slot[id(p)][1].data_ptr():
                        self._slot[id(p)][1].copy_(p.grad.reshape(-1))
                self._launch(bi)
        if self.comm is not None:
            for w in self._works:
                w.wait()
            torch.cuda.current_stream().wait_stream(self.comm)
        self._works = []
        stats = self._flat[0][-4:-2]
        den = (stats[1] + SMALL_POSIVITE_FLOAT) if self.denom == 'eps' else torch.clamp(stats[1], min=1.0)
        inv = 1.0 / den
        result = torch.stack([stats[0] * inv, stats[1]])
        for bi, flat in enumerate(self._flat):
            n = flat.numel() - (4 if bi == 0 else 0)
            flat[:n].mul_(inv)
        for p in self.params:
            p.grad = self._slot[id(p)][1].view(p.shape)
        return result[0], result[1]
