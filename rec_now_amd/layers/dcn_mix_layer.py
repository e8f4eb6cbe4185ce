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

    def _build_dnn_params(self, dim_in):
        N, S = self.num_expert, self.dim_sub_space
        self.origin_to_sub_kernels = [self.add_weight('origin_to_sub_kernels_of_layer%s' % l, shape=[N, dim_in, S],
                                                      initializer=self.kernel_initializer) for l in range(self.num_layer)]
        self.sub_to_sub_kernels = [self.add_weight('sub_to_sub_kernels_of_layer%s' % l, shape=[N, S, S],
                                                   initializer=self.kernel_initializer) for l in range(self.num_layer)]
        self.sub_to_origin_kernels = [self.add_weight('sub_to_origin_kernels_of_layer%s' % l, shape=[N, S, dim_in],
                                                      initializer=self.kernel_initializer) for l in range(self.num_layer)]
        self.biases = [self.add_weight('bias_of_layer%s' % l, shape=[1, N, dim_in], initializer=self.bias_initializer)
                       for l in range(self.num_layer)]

    def _build_gates(self, input_shape):
        from torch import nn
        self.gate_layers = nn.ModuleList([_GateDense(self.num_expert, 'gate_of_layer%s' % l) for l in range(self.num_layer)])
        for g in self.gate_layers:            # keras creates these kernels at first call; same RNG order (after the dnn params)
            g._build_device = self._build_device
            g.build(input_shape)

    def build(self, input_shape):
        dim_in = int(input_shape[-1])
        self._build_dnn_params(dim_in)
        self._build_gates(input_shape)
        self.built = True

    def call(self, inputs):
        """inputs (B, D) -> (B, D)."""
        params = (list(self.origin_to_sub_kernels) + list(self.sub_to_sub_kernels) + list(self.sub_to_origin_kernels)
                  + list(self.biases) + [g.kernel for g in self.gate_layers])
        return DCNMixFunction.apply(inputs, self.num_layer, self._act_inner, self._act_outer, *params)
