"""MMOELayer -- drop-in for rec_now/layers/mmoe_layer.py (/root/reference/rec_now/layers/mmoe_layer.py:14-126)."""
from torch import nn

from ._keras import DenseBase
from ._ops import moe_mix
from .multi_dense_layer import MultiDenseLayer


class MMOELayer(DenseBase):
    """Multi-gate Mixture-of-Experts.

    Symbols: B batch size, D input dim, N experts, T tasks, dim_out = dnn_dims[-1].
    """

    def __init__(self, num_task, num_experts, dnn_dims, **kwargs):
        super().__init__(0, **kwargs)
        self.num_task = num_task
        self.num_experts = num_experts
        self.dnn_dims = dnn_dims

    def _build_gates(self):
        name = f"{self.name}/gates"
        # reference :59: MultiDenseLayer(num_experts, num_task) with default Dense kwargs (bias, linear), then Softmax
        return MultiDenseLayer(self.num_experts, self.num_task, name=f'{name}/MultiDenseLayer')

    def _build_experts(self):
        name = f"{self.name}/experts"
        layers = []
        for layer_idx, dim in enumerate(self.dnn_dims):
            is_last_layer = layer_idx == len(self.dnn_dims) - 1
            activation = None if is_last_layer else self.activation
            layers.append(MultiDenseLayer(dim, self.num_experts, activation=activation,
                                          name=f'{name}/MultiDenseLayer_{layer_idx}', **self._dense_kwargs()))
        return nn.ModuleList(layers)

    def build(self, input_shape):
        self.dnn_experts = self._build_experts()
        self.gates = self._build_gates()
        self.built = True

    def call(self, inputs, merge_output=True):
        """inputs (B, D).  Returns (T, B, dim_out) if merge_output else a list of T tensors (B, dim_out)."""
        experts_output = inputs
        for layer in self.dnn_experts:
            experts_output = layer(experts_output)                      # (N, B, dim_out)
        gate_logits = self.gates(inputs)                                # (T, B, N); softmax is fused into the mix kernel
        output = moe_mix(gate_logits, list(experts_output.unbind(0)))   # (T, B, dim_out)
        if merge_output:
            return output
        return [output[task_idx] for task_idx in range(self.num_task)]
