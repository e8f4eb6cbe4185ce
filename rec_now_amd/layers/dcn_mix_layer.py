"""DCNMixLayer -- drop-in for rec_now/layers/dcn_mix_layer.py (/root/reference/rec_now/layers/dcn_mix_layer.py:12-151).
DCN-v2 mixture of low-rank experts; reference variant WITHOUT the residual term (:150)."""
from ._keras import Layer, activation_code
from ._ops import DCNMixFunction


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
        self._act_inner, cb_i = activation_code(activation_inner)
        self._act_outer, cb_o = activation_code(activation_outer)
        if cb_i is not None or cb_o is not None:
            raise NotImplementedError('DCNMixLayer fuses activations linear/relu/tanh/sigmoid only')

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
        params = (list(self.origin_to_sub_kernels) + list(self.sub_to_sub_kernels) + list(self.sub_to_origin_kernels)
                  + list(self.biases) + [g.kernel for g in self.gate_layers])
        return DCNMixFunction.apply(inputs, self.num_layer, self._act_inner, self._act_outer, *params)
