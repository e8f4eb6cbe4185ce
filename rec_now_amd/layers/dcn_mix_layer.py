"""DCNMixLayer -- drop-in for rec_now/layers/dcn_mix_layer.py (/root/reference/rec_now/layers/dcn_mix_layer.py:12-151).
DCN-v2 mixture of low-rank experts; reference variant WITHOUT the residual term (:150)."""
from ._keras import Layer, activation_code
from ._ops import DCNMixFunction, moe_mix, multi_dense


class _GateDense(Layer):
    """keras.layers.Dense(num_expert, use_bias=False, name='gate_of_layer{l}') of reference :98-103 (kernel only)."""

    def __init__(self, units, name):
        super().__init__(name=name)
        self.units = units

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', shape=[int(input_shape[-1]), self.units], initializer='glorot_uniform')
        self.built = True


class DCNMixLayer(Layer):
    """Symbols: B batch size, D input dim, S sub-space dim, N experts per layer, L layers."""

    def __init__(self, dim_sub_space, num_layer=1, num_expert=2,
                 activation_inner='tanh', activation_outer='tanh',
                 kernel_initializer='glorot_uniform', bias_initializer='zeros',
                 trainable=True, name=None, dtype=None, dynamic=False, **kwargs):
        super().__init__(trainable=trainable, name=name, dtype=dtype, dynamic=dynamic, **kwargs)
        self.dim_sub_space = dim_sub_space
        self.num_layer = num_layer
        self.num_expert = num_expert
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.activation_inner = activation_inner
        self.activation_outer = activation_outer
        # linear / relu / tanh / sigmoid are fused into the kernels; any other keras-style activation must be a callable on
        # torch tensors and selects the unfused route of `call` (reference :48-49 takes whatever keras.activations.get accepts)
        self._act_inner, self._cb_inner = activation_code(activation_inner)
        self._act_outer, self._cb_outer = activation_code(activation_outer)

    def build(self, input_shape):
        """Per layer l: origin_to_sub (N, D, S), sub_to_sub (N, S, S), sub_to_origin (N, S, D) kernels, bias (1, N, D) and the
        bias-free gate Dense (D, N) -- created family by family, gates last, which is the order Keras draws them in (the
        reference builds its gate layers lazily at the first call)."""
        from torch import nn
        D, N, S, L = int(input_shape[-1]), self.num_expert, self.dim_sub_space, self.num_layer
        family = lambda stem, shape, init: [self.add_weight('%s_of_layer%s' % (stem, l), shape=shape, initializer=init)   # noqa: E731
                                            for l in range(L)]
        self.origin_to_sub_kernels = family('origin_to_sub_kernels', [N, D, S], self.kernel_initializer)
        self.sub_to_sub_kernels = family('sub_to_sub_kernels', [N, S, S], self.kernel_initializer)
        self.sub_to_origin_kernels = family('sub_to_origin_kernels', [N, S, D], self.kernel_initializer)
        self.biases = family('bias', [1, N, D], self.bias_initializer)
        self.gate_layers = nn.ModuleList([_GateDense(N, 'gate_of_layer%s' % l) for l in range(L)])
        for gate in self.gate_layers:
            gate._build_device = self._build_device
            gate.build(input_shape)
        self.built = True

    def call(self, inputs):
        """inputs (B, D) -> (B, D)."""
        if self._cb_inner is not None or self._cb_outer is not None:
            return self._call_unfused(inputs)
        params = (list(self.origin_to_sub_kernels) + list(self.sub_to_sub_kernels) + list(self.sub_to_origin_kernels)
                  + list(self.biases) + [g.kernel for g in self.gate_layers])
        return DCNMixFunction.apply(inputs, self.num_layer, self._act_inner, self._act_outer, *params)

    def _call_unfused(self, inputs):
        """User-callable activations: the layer as the reference writes it (:135-150), one batched product per line on the
        MFMA GEMM (`multi_dense`), the softmax gate and the expert mix in the fused mix kernel, the callables in between."""
        N, D = self.num_expert, int(inputs.shape[-1])
        layer_input = inputs
        for l in range(self.num_layer):
            sub = multi_dense(layer_input, self.origin_to_sub_kernels[l], None, self._act_inner or 0)             # (N, B, S)  :135-136
            if self._cb_inner is not None:
                sub = self._cb_inner(sub)
            sub = multi_dense(sub, self.sub_to_sub_kernels[l], None, self._act_outer or 0)                         # (N, B, S)  :137-138
            if self._cb_outer is not None:
                sub = self._cb_outer(sub)
            origin = multi_dense(sub, self.sub_to_origin_kernels[l], self.biases[l].reshape(N, 1, D), 0)           # (N, B, D)  :141-142
            logits = multi_dense(layer_input, self.gate_layers[l].kernel.reshape(1, D, N), None, 0)[0]             # (B, N)     :146
            layer_input = inputs * moe_mix(logits, list(origin.unbind(0)))      # sum_n G_n (x * O_n) = x * sum_n G_n O_n     :143-150
        return layer_input
