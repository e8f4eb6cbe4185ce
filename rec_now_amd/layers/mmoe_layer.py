"""MMOELayer -- drop-in for rec_now/layers/mmoe_layer.py (/root/reference/rec_now/layers/mmoe_layer.py:14-126)."""
from torch import nn

from ._keras import DenseBase
from ._ops import moe_mix
from .multi_dense_layer import MultiDenseLayer


class MMOELayer(DenseBase):
    """Multi-gate Mixture-of-Experts: N expert DNNs (one batched MultiDenseLayer per DNN layer) shared by T tasks, each task
    mixing the expert outputs with its own softmax gate.

    Symbols: B batch size, D input dim, N experts, T tasks, dim_out = dnn_dims[-1].
    Variable names as the reference: `{name}/experts/MultiDenseLayer_{i}/...`, `{name}/gates/MultiDenseLayer/...`.
    """

    def __init__(self, num_task, num_experts, dnn_dims, **kwargs):
        super().__init__(0, **kwargs)
        self.num_task, self.num_experts, self.dnn_dims = num_task, num_experts, dnn_dims

    def build(self, input_shape):
        depth = len(self.dnn_dims)
        # experts: hidden layers carry the layer's activation, the last one is linear (reference :76-80)
        self.dnn_experts = nn.ModuleList([
            MultiDenseLayer(width, self.num_experts, activation=self.activation if i + 1 < depth else None,
                            name='%s/experts/MultiDenseLayer_%d' % (self.name, i), **self._dense_kwargs())
            for i, width in enumerate(self.dnn_dims)])
        # gates: T linear maps D -> N with default Dense settings (reference :59); their softmax is fused into the mix kernel
        self.gates = MultiDenseLayer(self.num_experts, self.num_task, name='%s/gates/MultiDenseLayer' % self.name)
        self.built = True

    def call(self, inputs, merge_output=True):
        """inputs (B, D).  Returns (T, B, dim_out) if merge_output else a list of T tensors (B, dim_out)."""
        hidden = inputs
        for layer in self.dnn_experts:
            hidden = layer(hidden)                                                  # (N, B, width)
        mixed = moe_mix(self.gates(inputs), list(hidden.unbind(0)))                 # gate logits (T, B, N) -> (T, B, dim_out)
        return mixed if merge_output else [mixed[t] for t in range(self.num_task)]
