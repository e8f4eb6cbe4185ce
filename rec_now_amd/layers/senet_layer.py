"""SENETLayer -- drop-in for rec_now/layers/senet_layer.py (/root/reference/rec_now/layers/senet_layer.py:14-119)."""
import torch

from .. import _lib
from ._keras import DenseBase, Layer, activation_code
from ._ops import multi_dense


class _Holder:
    """Per-call device arrays shared by the squeeze and scale nodes."""

    def __init__(self, xs):
        dev = xs[0].device
        self.F = len(xs)
        self.xs = xs
        dims = [int(x.shape[-1]) for x in xs]
        offs = [0]
        for d in dims[:-1]:
            offs.append(offs[-1] + d)
        self.total = sum(dims)
        self.dims_list = dims
        self.dims = _lib.const_array(dims, torch.int32, dev)
        self.offs = _lib.const_array(offs, torch.int32, dev)
        self.ptrs = _lib.ptr_array(xs, dev)
        # equal widths and 16-byte aligned rows: the float4 kernels (torch allocations are 256-byte aligned; views may not be)
        same = len(set(dims)) == 1 and dims[0] % 4 == 0 and all(x.data_ptr() % 16 == 0 for x in xs)
        self.uniform = dims[0] if same else 0


class _SenetFunction(torch.autograd.Function):
    """out = concat(fields) * w[:, field_of_column]  with  w = excite(mean_d fields)  as ONE autograd node: the excitation
    MLP runs inside forward under torch.enable_grad so that its own (MultiDense) backward can be replayed in backward,
    and the field gradient dx_f = dout * w + dsq / D_f is written in a single pass."""

    @staticmethod
    def forward(ctx, excite, n_params, *args):
        params, fields = args[:n_params], args[n_params:]
        xs = [_lib.f32c(x, 'SENET input') for x in fields]
        B = xs[0].shape[0]
        for x in xs:
            if x.dim() != 2 or x.shape[0] != B:
                raise ValueError('SENET inputs must be (B, D_f) tensors with one batch size')
        h = _Holder(xs)
        dev = xs[0].device
        sq = torch.empty((B, h.F), dtype=torch.float32, device=dev)
        _lib.call('recnow_senet_squeeze', _lib.ptr(h.ptrs), _lib.ptr(h.dims), _lib.ptr(h.offs), h.F, h.total, B, _lib.ptr(sq), h.uniform, _lib.stream())
        with torch.enable_grad():
            sq_leaf = sq.detach().requires_grad_(True)
            w = excite(sq_leaf)                                   # (B, F), the two Dense layers on the MFMA GEMM
        wd = _lib.f32c(w.detach(), 'excitation')
        out = torch.empty((B, h.total), dtype=torch.float32, device=dev)
        _lib.call('recnow_senet_scale_fwd', _lib.ptr(h.ptrs), _lib.ptr(h.dims), _lib.ptr(h.offs), h.F, h.total, B, _lib.ptr(wd),
                  _lib.ptr(out), h.uniform, _lib.stream())
        ctx.h, ctx.sq_leaf, ctx.w, ctx.wd, ctx.params = h, sq_leaf, w, wd, params
        return out

    @staticmethod
    def backward(ctx, dout):
        h, B = ctx.h, ctx.wd.shape[0]
        dev = ctx.wd.device
        dout = _lib.f32c(dout, 'grad')
        dw = torch.empty((B, h.F), dtype=torch.float32, device=dev)
        _lib.call('recnow_senet_scale_bwd_w', _lib.ptr(h.ptrs), _lib.ptr(h.dims), _lib.ptr(h.offs), h.F, h.total, B, _lib.ptr(dout),
                  _lib.ptr(dw), h.uniform, _lib.stream())
        live = [i for i, p in enumerate(ctx.params) if p.requires_grad]
        got = torch.autograd.grad(ctx.w, [ctx.sq_leaf] + [ctx.params[i] for i in live], dw, allow_unused=True)
        grads = [got[0]] + [None] * len(ctx.params)
        for k, i in enumerate(live):
            grads[1 + i] = got[1 + k]
        dsq = grads[0] if grads[0] is not None else torch.zeros_like(dw)
        dsq = _lib.f32c(dsq, 'grad')
        if h.uniform:                               # equal widths: one buffer, F blocks (one allocation instead of F)
            dx = torch.empty((h.F, B, h.uniform), dtype=torch.float32, device=dev)
            dxs, dptrs = dx.unbind(0), _lib.block_ptr_array(dx, h.F)
        else:
            dxs = [torch.empty_like(x) for x in h.xs]
            dptrs = _lib.ptr_array(dxs, dev)
        _lib.call('recnow_senet_scale_bwd_x', _lib.ptr(dptrs), _lib.ptr(h.dims), _lib.ptr(h.offs), h.F, h.total, B, _lib.ptr(ctx.wd),
                  _lib.ptr(dout), _lib.ptr(dsq), h.uniform, _lib.stream())
        return (None, None) + tuple(grads[1:]) + tuple(dxs)


class _SenetFused(torch.autograd.Function):
    """The whole layer as one kernel per direction (recnow_senet_fused_*): equal field widths, built-in activations, small
    F and hidden width.  Inputs: the two Dense kernels (+ biases or None), then the fields."""

    @staticmethod
    def forward(ctx, act1, act2, w1, b1, w2, b2, *fields):
        xs = [_lib.f32c(x, 'SENET input') for x in fields]
        B, D = xs[0].shape
        F, M = len(xs), w1.shape[1]
        dev = xs[0].device
        w1c, w2c = _lib.f32c(w1.detach(), 'kernel'), _lib.f32c(w2.detach(), 'kernel')
        b1c = _lib.f32c(b1.detach(), 'bias') if b1 is not None else None
        b2c = _lib.f32c(b2.detach(), 'bias') if b2 is not None else None
        out = torch.empty((B, F * D), dtype=torch.float32, device=dev)
        sq = torch.empty((B, F), dtype=torch.float32, device=dev)
        h = torch.empty((B, M), dtype=torch.float32, device=dev)
        w = torch.empty((B, F), dtype=torch.float32, device=dev)
        ptrs = _lib.ptr_array(xs, dev)
        _lib.call('recnow_senet_fused_fwd', _lib.ptr(ptrs), F, D, B, _lib.ptr(w1c), _lib.ptr(b1c), _lib.ptr(w2c), _lib.ptr(b2c), M,
                  act1, act2, _lib.ptr(out), _lib.ptr(sq), _lib.ptr(h), _lib.ptr(w), _lib.stream())
        ctx.save_for_backward(w1c, w2c, sq, h, w, *xs)
        ctx.meta = (act1, act2, b1 is not None, b2 is not None, ptrs)
        return out

    @staticmethod
    def backward(ctx, dout):
        w1c, w2c, sq, h, w, *xs = ctx.saved_tensors
        act1, act2, has_b1, has_b2, ptrs = ctx.meta
        B, D = xs[0].shape
        F, M = len(xs), w1c.shape[1]
        dev = dout.device
        dout = _lib.f32c(dout, 'grad')
        dx = torch.empty((F, B, D), dtype=torch.float32, device=dev)
        dw1, dw2 = torch.empty_like(w1c), torch.empty_like(w2c)
        db1 = torch.empty(M, dtype=torch.float32, device=dev) if has_b1 else None
        db2 = torch.empty(F, dtype=torch.float32, device=dev) if has_b2 else None
        ws = _lib.workspace(_lib.load().recnow_senet_fused_workspace_bytes(B, F, M), dev)
        _lib.call('recnow_senet_fused_bwd', _lib.ptr(ptrs), _lib.ptr(_lib.block_ptr_array(dx, F)), F, D, B, _lib.ptr(w1c), _lib.ptr(w2c), M,
                  act1, act2, _lib.ptr(dout), _lib.ptr(sq), _lib.ptr(h), _lib.ptr(w), _lib.ptr(dw1), _lib.ptr(db1), _lib.ptr(dw2),
                  _lib.ptr(db2), _lib.ptr(ws), ws.numel(), _lib.stream())
        return (None, None, dw1, db1, dw2, db2) + tuple(dx.unbind(0))


class _Dense(Layer):
    """keras.layers.Dense(units, activation, use_bias, ...) of the excitation MLP (:52-65) on recnow_multi_dense (N = 1)."""

    def __init__(self, units, activation, use_bias, kernel_initializer, bias_initializer, name):
        super().__init__(name=name)
        self.units, self.use_bias = units, use_bias
        self.act_code, self.act_callable = activation_code(activation)
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', shape=[int(input_shape[-1]), self.units], initializer=self.kernel_initializer)
        self.bias = self.add_weight('bias', shape=[self.units], initializer=self.bias_initializer) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        d, u = self.kernel.shape
        bias = self.bias.reshape(1, 1, u) if self.bias is not None else None
        y = multi_dense(inputs, self.kernel.reshape(1, d, u), bias, self.act_code if self.act_code is not None else 0)[0]
        return self.act_callable(y) if self.act_callable is not None else y


class SENETLayer(DenseBase):
    """Squeeze-Excitation network over field embeddings (FiBiNET); fields may have different widths.

    Symbols: B batch size, F fields, Df width of field f, total_dim = sum(Df).
    """

    def __init__(self, reduction_ratio, activation_inner='tanh', activation_outer='tanh', **kwargs):
        super().__init__(0, **kwargs)
        self.reduction_ratio = reduction_ratio
        self.activation_inner = activation_inner
        self.activation_outer = activation_outer

    def _get_middle_dim(self):
        return max(round(self.num_field * self.reduction_ratio), 1)       # :68-74

    def _build_senet(self):
        name = f"{self.name}/senet"
        dims = [self.middle_dim, self.num_field]
        acts = [self.activation_inner, self.activation_outer]
        return torch.nn.ModuleList([
            _Dense(dim, act, self.use_bias, self.kernel_initializer, self.bias_initializer, name=f'{name}/dense_{idx}')
            for idx, (dim, act) in enumerate(zip(dims, acts))])

    def build(self, input_shape):
        if not isinstance(input_shape, list):
            input_shape = [input_shape]
        self.num_field = len(input_shape)
        self.total_dim = sum(int(s[-1]) for s in input_shape)
        self.pos_idx = [f for f, s in enumerate(input_shape) for _ in range(int(s[-1]))]      # :84-88
        self.middle_dim = self._get_middle_dim()
        self.senet = self._build_senet()
        shape = (None, self.num_field)
        for layer in self.senet:                     # build eagerly so the parameters exist before the first call
            layer._build_device = self._build_device
            layer.build(shape)
            layer.built = True
            shape = (None, layer.units)
        self.built = True

    def _excite(self, sq):
        h = sq
        for layer in self.senet:
            h = layer(h)
        return h

    def call(self, inputs):
        """inputs: list of F tensors (B, Df) (a single tensor is wrapped, :101-102).  Returns (B, total_dim)."""
        if not isinstance(inputs, (list, tuple)):
            inputs = [inputs]
        fused = self._fused_plan(inputs)
        if fused is not None:
            d0, d1 = self.senet
            return _SenetFused.apply(fused[0], fused[1], d0.kernel, d0.bias, d1.kernel, d1.bias, *inputs)
        params = [p for layer in self.senet for p in layer.parameters()]
        return _SenetFunction.apply(self._excite, len(params), *params, *inputs)

    def _fused_plan(self, inputs):
        """(act1, act2) codes when the one-kernel-per-direction path applies: equal widths inside the kernel's limits,
        built-in activations, 16-byte aligned fp32 CUDA fields."""
        d0, d1 = self.senet
        if d0.act_code is None or d1.act_code is None:
            return None
        x0 = inputs[0]
        if not all(isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 2 and x.shape == x0.shape for x in inputs):
            return None
        if not _lib.load().recnow_senet_fused_supported(len(inputs), int(x0.shape[1]), int(d0.units)):
            return None
        return d0.act_code, d1.act_code
