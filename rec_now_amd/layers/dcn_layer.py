"""DCNLayer -- drop-in for rec_now/layers/dcn_layer.py (/root/reference/rec_now/layers/dcn_layer.py:12-103).
Reference variant of the DCN-v1 cross: x_{l+1} = activation(x0 * (x_l @ kernel_l) + bias_l), WITHOUT the `+ x_l`
residual of the paper (:95-98)."""
import torch

from ._keras import DenseBase
from ._ops import DCNFunction


class DCNLayer(DenseBase):
    """Cross network of Deep & Cross Network.  Symbols: B batch size, D input dim."""

    def __init__(self, degree_of_cross, **kwargs):
        """degree_of_cross: number of cross layers; other kwargs as keras.layers.Dense (activation, use_bias, ...)."""
        super().__init__(0, **kwargs)
        self.degree_of_cross = degree_of_cross
        if self.act_code is None:
            raise NotImplementedError('DCNLayer fuses activations linear/relu/tanh/sigmoid only')

    def _build_kernels(self):
        self.kernels = [self.add_weight(f'kernel_{layer_idx}', shape=[self.input_dim, 1],
                                        initializer=self.kernel_initializer, regularizer=self.kernel_regularizer,
                                        constraint=self.kernel_constraint, dtype=self.dtype, trainable=True)
                        for layer_idx in range(self.degree_of_cross)]

    def _build_biases(self):
        self.biases = [self.add_weight(f'bias_{layer_idx}', shape=[1, self.input_dim],
                                       initializer=self.bias_initializer, regularizer=self.bias_regularizer,
                                       constraint=self.bias_constraint, dtype=self.dtype, trainable=True)
                       for layer_idx in range(self.degree_of_cross)]

    def build(self, input_shape):
        self.input_dim = int(input_shape[-1])
        self._build_kernels()
        if self.use_bias:
            self._build_biases()
        else:
            self.biases = None
        self.built = True

    def call(self, inputs):
        """inputs (B, D) -> (B, D)."""
        D = self.input_dim
        kernels = torch.cat([k.reshape(1, D) for k in self.kernels], dim=0)           # (L, D)
        biases = torch.cat([b.reshape(1, D) for b in self.biases], dim=0) if self.use_bias else None
        return DCNFunction.apply(inputs, kernels, biases, self.act_code)
