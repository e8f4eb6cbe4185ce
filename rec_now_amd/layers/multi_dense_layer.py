"""MultiDenseLayer -- drop-in for rec_now/layers/multi_dense_layer.py
(/root/reference/rec_now/layers/multi_dense_layer.py:13-94): N same-shape Dense layers evaluated as one batched GEMM."""
import torch

from ._keras import DenseBase
from ._ops import multi_dense


class MultiDenseLayer(DenseBase):
    """N Dense layers with identical shapes.

    Input (B, D) (shared by all N) or (N, B, D); output (N, B, U).
    Symbols: B batch size, D input dim, N number of DNNs, U output dim.
    """

    def __init__(self, units, num_dnn, **kwargs):
        """units: output dim of each DNN; num_dnn: number of DNNs; other kwargs as keras.layers.Dense."""
        super().__init__(units, **kwargs)
        self.num_dnn = int(num_dnn)

    def build(self, input_shape):
        """Creates `kernel` (N, D, U) and, with use_bias, `bias` (N, 1, U) -- the reference's variable names and shapes."""
        if str(self.dtype) not in ('float32', 'torch.float32'):
            raise TypeError('Unable to build `MultiDenseLayer` layer with non-float32 dtype %s' % (self.dtype,))   # kernels are fp32
        width = input_shape[-1]
        if width is None:
            raise ValueError('The last dimension of the inputs to `Dense` should be defined. Found `None`.')
        n, u = self.num_dnn, self.units
        common = dict(dtype=self.dtype, trainable=True)
        self.kernel = self.add_weight('kernel', shape=[n, int(width), u], initializer=self.kernel_initializer,
                                      regularizer=self.kernel_regularizer, constraint=self.kernel_constraint, **common)
        self.bias = None
        if self.use_bias:
            self.bias = self.add_weight('bias', shape=[n, 1, u], initializer=self.bias_initializer,
                                        regularizer=self.bias_regularizer, constraint=self.bias_constraint, **common)
        self.built = True

    def call(self, inputs):
        """inputs: (B, D) or (N, B, D).  Returns (N, B, U)."""
        if inputs.dim() not in (2, 3):
            raise ValueError('MultiDenseLayer expects a (B, D) or (N, B, D) input, got shape %s' % (tuple(inputs.shape),))
        if inputs.dim() == 3 and inputs.shape[0] != self.num_dnn:
            if inputs.shape[0] == 1:
                inputs = inputs[0]
            else:
                # TF raises InvalidArgumentError from the batched matmul (tests/layers/test_multi_dense_layer.py:57-73)
                raise ValueError('In[0] and In[1] must have compatible batch dimensions: %s vs. %s'
                                 % (list(inputs.shape), list(self.kernel.shape)))
        if inputs.shape[-1] != self.kernel.shape[1]:
            raise ValueError('Matrix size-incompatible: In[0]: %s, In[1]: %s' % (list(inputs.shape), list(self.kernel.shape)))
        code = self.act_code if self.act_code is not None else 0
        out = multi_dense(inputs, self.kernel, self.bias, code)
        if self.act_callable is not None:
            out = self.act_callable(out)
        return out
