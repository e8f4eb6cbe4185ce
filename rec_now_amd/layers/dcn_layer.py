"""DCNLayer -- drop-in for rec_now/layers/dcn_layer.py (/root/reference/rec_now/layers/dcn_layer.py:12-103).
Reference variant of the DCN-v1 cross: x_{l+1} = activation(x0 * (x_l @ kernel_l) + bias_l), WITHOUT the `+ x_l`
residual of the paper (:95-98)."""
import torch

from ._keras import DenseBase
from ._ops import DCNFunction, DCNStepFunction


class DCNLayer(DenseBase):
    """Cross network of Deep & Cross Network.  Symbols: B batch size, D input dim."""

    def __init__(self, degree_of_cross, **kwargs):
        """degree_of_cross: number of cross layers; other kwargs as keras.layers.Dense (activation, use_bias, ...)."""
        super().__init__(0, **kwargs)
        self.degree_of_cross = degree_of_cross

    def build(self, input_shape):
        """kernel_l (D, 1) and, with use_bias, bias_l (1, D) for l < degree_of_cross (the reference's names and shapes)."""
        D = self.input_dim = int(input_shape[-1])
        make = lambda kind, l, shape, init, reg, con: self.add_weight(       # noqa: E731
            '%s_%d' % (kind, l), shape=shape, initializer=init, regularizer=reg, constraint=con, dtype=self.dtype, trainable=True)
        L = self.degree_of_cross
        self.kernels = [make('kernel', l, [D, 1], self.kernel_initializer, self.kernel_regularizer, self.kernel_constraint)
                        for l in range(L)]
        self.biases = ([make('bias', l, [1, D], self.bias_initializer, self.bias_regularizer, self.bias_constraint)
                        for l in range(L)] if self.use_bias else None)
        self.built = True

    def call(self, inputs):
        """inputs (B, D) -> (B, D)."""
        D = self.input_dim
        if self.act_callable is not None:
            # a user callable as activation (keras.activations.get accepts any): one cross-layer kernel per layer, the callable
            # runs between them on torch tensors (its gradient is torch autograd's)
            layer_input = inputs
            for l in range(self.degree_of_cross):
                bias = self.biases[l].reshape(D) if self.use_bias else None
                layer_input = self.act_callable(DCNStepFunction.apply(inputs, layer_input, self.kernels[l].reshape(D), bias, 0))
            return layer_input
        kernels = torch.cat([k.reshape(1, D) for k in self.kernels], dim=0)           # (L, D)
        biases = torch.cat([b.reshape(1, D) for b in self.biases], dim=0) if self.use_bias else None
        return DCNFunction.apply(inputs, kernels, biases, self.act_code)
