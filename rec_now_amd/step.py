"""The north-star training step as phases of ONE C entry point (`recnow_dcn_mix_step`, include/recnow.h):

    x -> DCNMixLayer -> MultiDenseLayer(1, 1) head -> pairwise_loss(scores, labels, groups) -> every gradient

i.e. /root/reference/rec_now/layers/dcn_mix_layer.py:114-151 -> multi_dense_layer.py:80-94 ->
rec_block/pairwise_loss_from_batch.py:228-279 and their backward, on buffers that are allocated ONCE.  Why it exists: under
strong scaling the metric's batch of 65 536 rows leaves 8192 rows per GPU at 8 GPUs, a step of ~0.5 ms of GPU time made of ~60
launches -- less than the host needs to enqueue them through autograd (0.6-0.9 ms).  Here a phase is one ctypes call, nothing is
allocated and no library event is recorded inside a phase, so each phase is also capturable into a HIP graph
(`capture()` / `replay()`): the host side of a step is then a handful of graph launches.

The two drop-in layers keep holding the weights (reference names); gradients land in `p.grad` exactly as after
`pairwise_loss(head(cross(x)), labels, groups).backward()` (same kernels; tests/test_step_gpu.py holds the two routes together).

Data parallel (rec_now_amd/dp.py): pass the `LayerwiseReducer`; the gradients are then written straight into its per-layer
buckets, the loss statistics into the first bucket's tail, the backward pass is cut into one piece per cross layer and each
bucket's all-reduce is launched as soon as its piece has been enqueued (the collectives are NOT captured: RCCL launches stay
ordinary stream work between the graph launches).
"""
import ctypes
import os

import torch

from . import _lib
from .fused import GpuEvent, fused_route_available, score_params
from .layers._ops import _host_ptr_array
from .rec_block._segments import _as_key_tensor

_GROUP, _FORWARD, _LOSS, _BACKWARD = 1, 2, 4, 8


class DCNMixPairwiseStep(object):
    """cross: DCNMixLayer (built), head: MultiDenseLayer(1, 1) (built, linear), both on the GPU.

    x (B, D) float32, labels (B,), groups (B,) float32 / int32 / float64 / int64 ids, optional mask (B,) bool: device tensors
    whose STORAGE is reused by every step (copy new batches into them) -- a captured graph replays addresses.
    RAGGED batches (B not a multiple of 256: what `dp.shard_rows_by_group` hands a rank, whole groups per rank): the step owns padded
    storage of B_pad = ceil(B / 256) * 256 rows behind `self.x`, `self.scores`, `self.dx` (each a view of the first B rows; `x` is
    COPIED into it once -- refill `step.x`, not the tensor passed in); the padding rows of x are zero and stay zero, the layers run
    over B_pad rows on the same kernels as any other batch, the grouping and the loss over the B rows of the batch, and the padding
    rows receive d loss / d score = 0, i.e. they add exactly nothing to any gradient (recnow_dcn_mix_step_desc.B_pad).
    reduce_mean: the reference's mean over the pairs (`loss` = sum / (P + 1e-10)); False: the loss sum and its gradient (the
    data-parallel form: the reducer divides by the global pair count).  need_dx: also d loss / d x (`self.dx`).
    reducer: optional dp.LayerwiseReducer built over `stages_for(cross, head)`.
    """

    def __init__(self, cross, head, x, labels, groups, mask=None, reduce_mean=True, need_dx=True, factor=1.0,
                 only_use_wrong_order_pair=False, reducer=None, two_streams=False):
        B = int(x.shape[0]) if x.dim() == 2 else 0
        B_pad = -(-B // 256) * 256
        if B < 1 or not fused_route_available(cross, head, x, rows=B_pad):
            raise ValueError('DCNMixPairwiseStep needs the fused north-star shape (recnow_dcn_mix_score_supported): N*S and D multiples '
                             'of 128, built-in activations, a linear MultiDenseLayer(1, 1) head (any B >= 1: ragged batches run on padded storage)')
        self.cross, self.head, self.reducer = cross, head, reducer
        if reducer is not None:
            reduce_mean = False        # the data-parallel form: loss SUM and its gradient, divided by the GLOBAL pair count by the reducer
        dev = x.device
        self.device = dev
        N, D, S = cross.origin_to_sub_kernels[0].shape
        L = cross.num_layer
        self.B, self.B_pad, self.D, self.L = B, B_pad, D, L
        if B_pad == B:
            self.x = _lib.f32c(x, 'inputs')
            self._x_store = self.x
        else:                              # ragged: padded storage owned by the step, the padding rows zero once and for all
            self._x_store = torch.zeros((B_pad, D), dtype=torch.float32, device=dev)
            self.x = self._x_store[:B]
            self.x.copy_(_lib.f32c(x, 'inputs'))
        self.labels = _lib.f32c(labels, 'labels').reshape(-1)
        self.groups, gdt = _as_key_tensor(groups)
        self.mask = None if mask is None else (mask.reshape(-1) != 0).to(torch.uint8).contiguous()
        if self.labels.numel() != B or self.groups.numel() != B or (self.mask is not None and self.mask.numel() != B):
            raise ValueError('labels / groups / mask must have %d elements' % B)
        lib = _lib.load()
        self._scores_store = torch.empty(B_pad, dtype=torch.float32, device=dev)
        self.scores = self._scores_store[:B]
        self.loss = torch.empty((), dtype=torch.float32, device=dev)
        self.n_pair = torch.empty(1, dtype=torch.int64, device=dev)
        self._dx_store = torch.empty((B_pad, D), dtype=torch.float32, device=dev) if need_dx else None
        self.dx = self._dx_store[:B] if need_dx else None
        self.ws = _lib.workspace(lib.recnow_dcn_mix_step_workspace_bytes(B_pad, D, S, N, L, gdt), dev)
        # parameters in the order of fused.score_params: head kernel, head bias, U_0.., V_0.., W_0.., bias_0.., gate_0..
        self.params = score_params(cross, head)
        self.grads = []
        for p in self.params:
            buf = reducer.buffer_of(p) if reducer is not None else None
            self.grads.append(torch.empty(p.numel(), dtype=torch.float32, device=dev) if buf is None else buf)
        self.stats = reducer.stats_slot() if reducer is not None else torch.empty(2, dtype=torch.float32, device=dev)
        ps = [_lib.f32c(p.detach(), 'weight') for p in self.params]
        for p, q in zip(self.params, ps):
            if p.data_ptr() != q.data_ptr():
                raise ValueError('the weights must be contiguous float32 tensors (the step reads them in place)')
        g5 = lambda i: self.grads[2 + i * L:2 + (i + 1) * L]        # noqa: E731
        w5 = lambda i: ps[2 + i * L:2 + (i + 1) * L]                # noqa: E731
        self._keep = [_host_ptr_array(w5(i)) for i in range(5)] + [_host_ptr_array(g5(i)) for i in range(5)]      # host pointer arrays: alive as long as the descriptor
        d = _lib.StepDesc()
        d.B, d.D, d.S, d.N, d.L = B, D, S, N, L
        d.B_pad = B_pad if B_pad != B else 0
        d.act_inner, d.act_outer, d.group_dtype = cross._act_inner, cross._act_outer, gdt
        d.only_use_wrong_order_pair, d.reduce_mean, d.factor = int(bool(only_use_wrong_order_pair)), int(bool(reduce_mean)), float(factor)
        P = lambda t: None if t is None else t.data_ptr()           # noqa: E731
        d.x, d.labels, d.groups, d.mask = P(self._x_store), P(self.labels), P(self.groups), P(self.mask)
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)            # noqa: E731
        d.U_host, d.V_host, d.W_host, d.bias_host, d.gate_host = (cast(a) for a in self._keep[:5])
        d.dU_host, d.dV_host, d.dW_host, d.dbias_host, d.dgate_host = (cast(a) for a in self._keep[5:])
        d.head_w, d.head_b = P(ps[0]), P(ps[1])
        d.dhead_w, d.dhead_b = P(self.grads[0]), P(self.grads[1])
        d.scores, d.loss, d.n_pair, d.stats, d.dx = P(self._scores_store), P(self.loss), P(self.n_pair), P(self.stats), P(self._dx_store)
        d.ws, d.ws_bytes = P(self.ws), self.ws.numel()
        # optional second stream of the backward pass (weight-gradient products beside the data-gradient chain): pays at small
        # shards, where a single product leaves most of the chip idle (8192 rows: -12 % per step, eager); neutral at 65 536 rows
        self.side2 = torch.cuda.Stream(device=dev) if two_streams else None
        d.stream2 = None if self.side2 is None else self.side2.cuda_stream
        self.desc = d
        self.reduce_mean = bool(reduce_mean)
        self.side = torch.cuda.Stream(device=dev)
        self._grouped = GpuEvent()
        self._fork = GpuEvent()
        self._graphs = None
        # the layer pieces of the backward pass: one per reducer stage (top layer + head first), or the whole pass
        self.pieces = [(L - 1 - i, L - 1 - i) for i in range(L)] if reducer is not None else [(L - 1, 0)]
        # stage i of the reducer = cross layer L-1-i: its event is what the library records for that layer (eager steps)
        self._layer_events = None
        if reducer is not None:
            self._layer_events = (ctypes.c_void_p * L)(*[reducer.events[L - 1 - l].handle for l in range(L)])
        self._shape = (B_pad, D, S, N, L)
        self._bind_grads()                     # p.grad = the step's gradient storage, written in place by every step

    def tile_route(self):
        """Whether THIS call's cross layers run the row-block persistent kernels: the library's own rule (csrc/dcnmix.hip `mix_tile_on` through
        recnow_dcn_mix_tile_route: shape, batch, RECNOW_TILE and the precision mode, read per call) -- asked, not restated."""
        return _lib.load().recnow_dcn_mix_tile_route(*self._shape) == 1

    def route_code(self):
        """recnow_dcn_mix_tile_route of THIS call: 0 one launch per product, 1 row-block kernels in both directions, 2 split-precision row-block forward
        (csrc/dcnmix_tile_split.hip) in front of the launch-per-product backward."""
        return int(_lib.load().recnow_dcn_mix_tile_route(*self._shape))

    @staticmethod
    def stages_for(cross, head):
        """Parameter stages for dp.LayerwiseReducer, in the order their gradients become final (top cross layer + head first)."""
        L = cross.num_layer
        per = lambda l: [cross.origin_to_sub_kernels[l], cross.sub_to_sub_kernels[l], cross.sub_to_origin_kernels[l], cross.biases[l],      # noqa: E731
                         cross.gate_layers[l].kernel]
        return [per(L - 1) + [head.kernel, head.bias]] + [per(l) for l in range(L - 2, -1, -1)]

    # ---- phases -------------------------------------------------------------------------------------------------------
    def _call(self, phases, hi=-1, lo=0, stream=None):
        _lib.call('recnow_dcn_mix_step', ctypes.byref(self.desc), phases, hi, lo, _lib.stream() if stream is None else _lib._P(stream.cuda_stream))

    def _bind_grads(self):
        """`p.grad` of every parameter is the step's gradient storage (re-bound after `zero_grad(set_to_none=True)` or `p.grad = None`)."""
        for p, g in zip(self.params, self.grads):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g.view(p.shape)

    def _enqueue(self, launch, whole_backward=False):
        """One step on the current stream (+ the side stream for the grouping).  `launch(key, fn)` runs piece `key` (eager: calls fn;
        replay: launches its graph).  Returns after everything is enqueued."""
        self._bind_grads()
        main = torch.cuda.current_stream()
        # Where the grouping of the batch runs (A/B switch RECNOW_STEP_GROUP = side | inline | after).  Under the launch-per-product forward it
        # hides on a side stream.  The row-block forward (csrc/dcnmix_tile.hip, batches <= 16 384 rows) holds every CU by itself: the
        # grouping's one workgroup only gets a CU when that launch drains, and the fork / join events are pure cost -- measured 0.676-0.687
        # (side) vs 0.667-0.672 ms (inline) per step at 8192 rows, 1.105 vs 1.087-1.091 ms at 16 384.
        # Round 5, 65 536 rows (tools/ab_env.sh, one box, two alternating repetitions): side 3.362 / 3.391, inline 3.350 / 3.351, after 3.362 / 3.370 ms per step.
        # Since the integer-image sort (round 4) the grouping launch takes 42-50 us by itself; under the forward GEMMs it was stretched to 160-280 us
        # of shared CUs and bought nothing back: inline everywhere (RECNOW_STEP_GROUP=side keeps the side-stream placement for A/B).
        mode = os.environ.get('RECNOW_STEP_GROUP') or 'inline'
        self.group_mode = mode
        if mode == 'side':
            # grouping (sort by group id, segments) does not depend on the scores: on a side stream, under the forward pass
            _lib.call('recnow_event_record', self._fork.handle, _lib._P(main.cuda_stream))
            self._fork.wait(self.side)
            torch.cuda.set_stream(self.side)           # (the `with torch.cuda.stream(..)` form costs ~40 us of host time)
            try:
                launch('group', lambda: self._call(_GROUP))
            finally:
                torch.cuda.set_stream(main)
            _lib.call('recnow_event_record', self._grouped.handle, _lib._P(self.side.cuda_stream))
            launch('forward', lambda: self._call(_FORWARD))
            self._grouped.wait(main)
        elif mode == 'inline':
            # ONE call for both phases: at shard sizes the library then packs the row-block kernels' weights on spare workgroups of the grouping
            # launch (k_front_small / k_front_mid, csrc/scan_sort.hip) instead of in a launch of its own in front of the forward kernel
            # ... and, with the loss stage in the same call, clears the pair counter there, so that the pair walk (which fills its LDS stages
            # straight from scores / labels / mask through the sorted order) is the loss stage's first launch: no pack kernel, no fill
            launch('front', lambda: self._call(_GROUP | _FORWARD | _LOSS))
        else:
            launch('forward', lambda: self._call(_FORWARD))
            launch('group', lambda: self._call(_GROUP))
        loss0 = 0 if mode == 'inline' else _LOSS      # (inline: the loss stage ran in the front call)
        if whole_backward:
            # eager under a reducer: ONE call walks all layers (the weight-gradient products of a layer run on the second stream beside
            # the chain of the layers below) and records the stages' events itself where each stage's last gradient is issued
            self.desc.layer_events_host = ctypes.cast(self._layer_events, ctypes.c_void_p)
            try:
                self._call(loss0 | _BACKWARD, self.L - 1, 0)
            finally:
                self.desc.layer_events_host = None
            for i in range(len(self.pieces)):
                self.reducer.stage_done(i, recorded=True)
            return self.reducer.reduce_in_place()
        for i, (hi, lo) in enumerate(self.pieces):
            launch(('bwd%d' if loss0 else 'bwdx%d') % i, lambda hi=hi, lo=lo, i=i: self._call((loss0 if i == 0 else 0) | _BACKWARD, hi, lo))
            if self.reducer is not None:
                self.reducer.stage_done(i)
        if self.reducer is not None:
            return self.reducer.reduce_in_place()
        return self.loss, self.n_pair

    def run(self):
        """One eager step.  Returns (loss, n_pair): 0-dim / 1-element device tensors (global mean loss and pair count under a
        reducer).  Gradients are in `p.grad` of every parameter, d loss / d x in `self.dx`.

        ALIASING: the returned tensors, `self.scores`, `self.dx` and every `p.grad` are views of storage that the NEXT step overwrites
        (that is what makes the step allocation-free and capturable).  Keep a value across steps with `.clone()` (8 bytes for the loss)."""
        return self._enqueue(lambda key, fn: fn(), whole_backward=self.reducer is not None)

    def capture(self):
        """Capture every piece of the step into its own HIP graph (single-stream graphs: a forked capture replays with a host-side
        join between the branches, +1.5 ms per step measured).  Call once, after a warm-up `run()`.

        REQUIRES `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` in the environment BEFORE `import torch` (the HIP runtime reads it when it loads): with
        ROCm 7.0's graph AQL-packet capture on, replaying these graphs after the same kernels were launched eagerly in between (a `run()`
        between two `replay()`s) ends in a GPU memory access fault that takes the process down (tools/graph_dist_probe.py).
        Raises RuntimeError when the switch is not set, instead of capturing graphs that may fault later."""
        import os
        if os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') != '0':
            raise RuntimeError("DCNMixPairwiseStep.capture(): export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before `import torch` (ROCm 7.0's graph "
                               'packet capture faults when captured kernels are also launched eagerly between replays; see INTEGRATION.md)')
        torch.cuda.synchronize(self.device)
        self._graphs = {}
        cap = torch.cuda.Stream(device=self.device)
        keys = [('group', lambda: self._call(_GROUP)), ('forward', lambda: self._call(_FORWARD)), ('front', lambda: self._call(_GROUP | _FORWARD | _LOSS))]
        for i, (hi, lo) in enumerate(self.pieces):
            keys.append(('bwd%d' % i, lambda hi=hi, lo=lo, i=i: self._call((_LOSS if i == 0 else 0) | _BACKWARD, hi, lo)))
            keys.append(('bwdx%d' % i, lambda hi=hi, lo=lo: self._call(_BACKWARD, hi, lo)))      # behind a front piece that holds the loss stage
        stream2, self.desc.stream2 = self.desc.stream2, None      # captured pieces are single-stream (a forked capture replays slowly)
        try:
            with torch.cuda.stream(cap):
                for key, fn in keys:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: under a process group the RCCL watchdog thread polls its events while this thread captures; in the default
                    # 'global' mode that poll is an illegal call DURING A CAPTURE and takes the process down (seen once the eager step before
                    # capture() got short enough for a collective's bookkeeping to be still pending)
                    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
                        fn()
                    self._graphs[key] = g
        finally:
            self.desc.stream2 = stream2
        torch.cuda.synchronize(self.device)
        return self

    def replay(self):
        """One step from the captured graphs (same results as `run()`, bit for bit; the same aliasing: clone what must outlive the next step)."""
        if self._graphs is None:
            raise RuntimeError('capture() first')
        return self._enqueue(lambda key, fn: self._graphs[key].replay())
