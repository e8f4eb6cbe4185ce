"""PLELayer -- drop-in for rec_now/layers/ple_layer.py (/root/reference/rec_now/layers/ple_layer.py:16-321).
Progressive Layered Extraction: per layer, shared and task-specific expert groups (MultiDenseLayer stacks) mixed by
softmax gates.  The control flow below follows the reference method by method; the compute is the batched fp32 MFMA
GEMM (experts, gates) and the fused softmax-mix kernel (no concat of expert outputs is materialised)."""
import copy

import torch
from torch import nn

from ._keras import DenseBase, Layer
from ._ops import moe_mix, multi_dense
from .multi_dense_layer import MultiDenseLayer


class _GateDense(Layer):
    """keras.layers.Dense(units) (with bias) of reference :170; the Softmax of :172 is fused into the mix kernel."""

    def __init__(self, units, name):
        super().__init__(name=name)
        self.units = units

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', shape=[int(input_shape[-1]), self.units], initializer='glorot_uniform')
        self.bias = self.add_weight('bias', shape=[self.units], initializer='zeros')
        self.built = True

    def call(self, inputs):
        d, u = self.kernel.shape
        return multi_dense(inputs, self.kernel.reshape(1, d, u), self.bias.reshape(1, 1, u), 0)[0]      # (B, units) logits


class _ExpertDNN(Layer):
    """keras Sequential of MultiDenseLayers (reference :125-155)."""

    def __init__(self, layers, name):
        super().__init__(name=name)
        self.layers_ = nn.ModuleList(layers)

    def call(self, inputs):
        h = inputs
        for layer in self.layers_:
            h = layer(h)
        return h


class PLELayer(DenseBase):
    """Symbols: B batch size, D input dim, N experts relevant to a task at a layer, dim_out DNN output dim."""

    def __init__(self, num_task, list_of_dnn_dims, list_of_num_experts_per_task, num_shared_task=1, **kwargs):
        if not isinstance(list_of_dnn_dims, list):
            raise TypeError('`list_of_dnn_dims` must be a list or list[list]')
        super().__init__(0, **kwargs)
        self.num_task = num_task
        self.num_shared_task = num_shared_task
        self.num_total_task = num_task + num_shared_task
        if num_shared_task > num_task:
            # reference :108 builds the shared names with range(num_task); every zip() then silently drops task groups
            # (SURVEY Appendix B15).  Refuse instead of reproducing the truncation.
            raise ValueError('num_shared_task > num_task is not supported (the reference silently drops task groups)')
        self.list_of_dnn_dims, self.list_of_num_experts_per_task, self.is_shared_tasks, self.task_names = \
            self._get_normalized_params(num_task, num_shared_task, list_of_dnn_dims, list_of_num_experts_per_task)

    @classmethod
    def _extend_int_list(cls, list_or_int, size_extend):
        if not isinstance(list_or_int, (int, list)):
            raise TypeError('`list_or_int` must be of type `int` or `list of int`, but got `%s`' % type(list_or_int))
        if isinstance(list_or_int, int):
            list_or_int = [list_or_int]
        if not list_or_int:
            raise ValueError('list can not be empty')
        list_or_int = copy.copy(list_or_int)
        while len(list_or_int) < size_extend:
            list_or_int.append(list_or_int[-1])
        return list_or_int

    @classmethod
    def _get_normalized_params(cls, num_task, num_shared_task, list_of_dnn_dims, list_of_num_experts_per_task):
        num_total_task = num_task + num_shared_task
        num_layer = len(list_of_dnn_dims)
        list_of_num_experts_per_task = cls._extend_int_list(list_of_num_experts_per_task, num_layer)
        list_of_num_experts_per_task = [cls._extend_int_list(n, num_total_task) for n in list_of_num_experts_per_task]
        list_of_dnn_dims = [cls._extend_int_list(dim, 1) for dim in list_of_dnn_dims]
        is_shared_tasks = [True] * num_shared_task
        task_names = [f'shared_{task_idx}' for task_idx in range(num_task)]       # sic (reference :108)
        for task_idx in range(num_task):
            is_shared_tasks.append(False)
            task_names.append(f'special_{task_idx}')
        return list_of_dnn_dims, list_of_num_experts_per_task, is_shared_tasks, task_names

    def _get_dnn_name(self, layer_name, task_name):
        return f'{self.name}/ple_layer_{layer_name}/task_{task_name}'

    def _get_gate_name(self, layer_name, task_name):
        return f'{self.name}/ple_gate_{layer_name}/task_{task_name}'

    def _build_one_dnn(self, dnn_dims, num_experts, layer_name, task_name):
        name = self._get_dnn_name(layer_name, task_name)
        layers = []
        for idx, dim in enumerate(dnn_dims):
            is_last_layer = idx == len(dnn_dims) - 1
            activation = None if is_last_layer else self.activation
            layers.append(MultiDenseLayer(dim, num_experts, activation=activation, name=f'{name}/MultiDenseLayer_{idx}',
                                          **self._dense_kwargs()))
        return _ExpertDNN(layers, name)

    def _build_one_gate(self, units, layer_name, task_name):
        name = self._get_gate_name(layer_name, task_name)
        return _GateDense(units, f'{name}/dense')

    def _get_experts_num(self, num_experts_per_task):
        num_total_experts = 0
        num_shared_experts = 0
        for is_shared_task, num_experts in zip(self.is_shared_tasks, num_experts_per_task):
            num_total_experts += num_experts
            if is_shared_task:
                num_shared_experts += num_experts
        return num_total_experts, num_shared_experts

    def _build_one_layer(self, layer_idx, dnn_dims, num_experts_per_task, num_layer):
        is_last_layer = layer_idx == num_layer - 1
        layer_dnns = []
        layer_gates = []
        num_total_experts, num_shared_experts = self._get_experts_num(num_experts_per_task)
        task_params = zip(self.is_shared_tasks, self.task_names, num_experts_per_task)
        for is_shared_task, task_name, num_experts in task_params:
            layer_dnns.append(self._build_one_dnn(dnn_dims, num_experts, layer_idx, task_name))
            if is_shared_task and is_last_layer:
                layer_gates.append(None)
            else:
                gate_output_dim = num_experts + num_shared_experts if not is_shared_task else num_total_experts
                layer_gates.append(self._build_one_gate(gate_output_dim, layer_idx, task_name))
        return layer_dnns, layer_gates

    def build(self, input_shape):
        num_layer = len(self.list_of_dnn_dims)
        self.dnns = []
        self.gates = []
        layer_params = zip(range(num_layer), self.list_of_dnn_dims, self.list_of_num_experts_per_task)
        for layer_idx, dnn_dims, num_experts_per_task in layer_params:
            layer_dnns, layer_gates = self._build_one_layer(layer_idx, dnn_dims, num_experts_per_task, num_layer)
            self.dnns.append(layer_dnns)
            self.gates.append(layer_gates)
        # register for nn.Module bookkeeping (parameters(), .to(), state_dict())
        self._dnn_modules = nn.ModuleList([m for layer in self.dnns for m in layer])
        self._gate_modules = nn.ModuleList([m for layer in self.gates for m in layer if m is not None])
        self.built = True

    def _get_input(self, last_layer_outputs, task_idx, is_shared_task):
        if is_shared_task:
            return torch.cat(last_layer_outputs, dim=-1)
        inputs = [last_layer_outputs[task_idx]]
        for last_output, shared in zip(last_layer_outputs, self.is_shared_tasks):
            if shared:
                inputs.append(last_output)
        return torch.cat(inputs, dim=-1)

    def _get_gate_input(self, dnn_outputs, task_idx, is_shared_task):
        """The experts a gate mixes, as a list of (B, dim_out) tensors in the reference's concat order (:238-257):
        all groups for a shared task; own group then the shared groups for a specific task.  No concat is made."""
        if is_shared_task:
            groups = list(dnn_outputs)
        else:
            groups = [dnn_outputs[task_idx]]
            for last_output, shared in zip(dnn_outputs, self.is_shared_tasks):
                if shared:
                    groups.append(last_output)
        experts = []
        for g in groups:
            experts.extend(g.unbind(0))
        return experts

    def _apply_dnns(self, is_first_layer, inputs, outputs, layer_dnns):
        dnn_outputs = []
        task_inputs = []
        dnn_params = zip(range(self.num_total_task), self.is_shared_tasks, layer_dnns)
        for task_idx, is_shared_task, dnn in dnn_params:
            last_layer_outputs = None if is_first_layer else outputs[-1]
            dnn_input = inputs if is_first_layer else self._get_input(last_layer_outputs, task_idx, is_shared_task)
            task_inputs.append(dnn_input)
            dnn_outputs.append(dnn(dnn_input))            # (N, B, dim_out)
        return dnn_outputs, task_inputs

    def _apply_gates(self, is_last_layer, dnn_outputs, task_inputs, layer_gates):
        gated_outputs = []
        gate_params = zip(range(self.num_total_task), self.is_shared_tasks, layer_gates)
        for task_idx, is_shared_task, gate in gate_params:
            if is_shared_task and is_last_layer:
                gated_outputs.append(None)
                continue
            gate_logits = gate(task_inputs[task_idx])                                   # (B, N)
            experts = self._get_gate_input(dnn_outputs, task_idx, is_shared_task)       # N x (B, dim_out)
            gated_outputs.append(moe_mix(gate_logits, experts))                         # (B, dim_out)
        return gated_outputs

    def call(self, inputs):
        """inputs (B, D).  Returns the num_task task outputs (shared tasks excluded), each (B, dim_out)."""
        num_layer = len(self.list_of_dnn_dims)
        outputs = []
        for layer_idx in range(num_layer):
            is_first_layer = layer_idx == 0
            is_last_layer = layer_idx == num_layer - 1
            dnn_outputs, task_inputs = self._apply_dnns(is_first_layer, inputs, outputs, self.dnns[layer_idx])
            gated_outputs = self._apply_gates(is_last_layer, dnn_outputs, task_inputs, self.gates[layer_idx])
            outputs.append(gated_outputs)
        return [output for output in outputs[-1] if output is not None]
