"""PLELayer -- drop-in for rec_now/layers/ple_layer.py (/root/reference/rec_now/layers/ple_layer.py:16-321).
Progressive Layered Extraction: per layer, shared and task-specific expert groups (MultiDenseLayer stacks) mixed by
softmax gates.  Same constructor, weights (by Keras name) and outputs as the reference; the wiring is a plan resolved at
build time, the compute is the batched fp32 MFMA GEMM (experts, gates) and the fused softmax-mix kernel (no concat of
expert outputs is materialised)."""
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
    """Progressive Layered Extraction.  Task groups = `num_shared_task` shared groups followed by `num_task` task-specific
    ones; per PLE layer every group owns a stack of batched expert DNNs and (except the shared groups of the last layer) a
    softmax gate over the experts it may see: a specific group sees its own experts and the shared ones, a shared group
    sees all of them.

    Symbols: B batch size, D input dim, N experts visible to a group at a layer, dim_out last DNN width of a layer.

    The wiring is resolved once in `build` into a plan (which previous outputs feed a group, which expert groups its gate
    mixes), `call` just walks the plan; the variable names are the reference's
    (`{name}/ple_layer_{l}/task_{group}/.../MultiDenseLayer_{i}/kernel`, `{name}/ple_gate_{l}/task_{group}/dense/kernel`).
    """

    def __init__(self, num_task, list_of_dnn_dims, list_of_num_experts_per_task, num_shared_task=1, **kwargs):
        if not isinstance(list_of_dnn_dims, list):
            raise TypeError('`list_of_dnn_dims` must be a list or list[list]')                  # reference :42-43
        super().__init__(0, **kwargs)
        if num_shared_task > num_task:
            # reference :108 names the shared groups with range(num_task); its zip()s then silently drop task groups
            # (SURVEY Appendix B15).  Refuse instead of reproducing the truncation.
            raise ValueError('num_shared_task > num_task is not supported (the reference silently drops task groups)')
        self.num_task, self.num_shared_task = num_task, num_shared_task
        self.num_total_task = num_task + num_shared_task
        dims, experts, shared, names = self._get_normalized_params(num_task, num_shared_task, list_of_dnn_dims,
                                                                   list_of_num_experts_per_task)
        self.list_of_dnn_dims, self.list_of_num_experts_per_task = dims, experts
        self.is_shared_tasks, self.task_names = shared, names

    # -- argument normalisation (class methods of the reference, :63-113: the reference's tests call them) ------------
    @classmethod
    def _extend_int_list(cls, list_or_int, size_extend):
        """int or list of int -> list of at least `size_extend` ints, padded with its last element."""
        if isinstance(list_or_int, int):
            values = [list_or_int]
        elif isinstance(list_or_int, list):
            values = list(list_or_int)
        else:
            raise TypeError('`list_or_int` must be of type `int` or `list of int`, but got `%s`' % type(list_or_int))
        if len(values) == 0:
            raise ValueError('list can not be empty')
        return values + [values[-1]] * max(0, size_extend - len(values))

    @classmethod
    def _get_normalized_params(cls, num_task, num_shared_task, list_of_dnn_dims, list_of_num_experts_per_task):
        n_groups, n_layers = num_task + num_shared_task, len(list_of_dnn_dims)
        per_layer = cls._extend_int_list(list_of_num_experts_per_task, n_layers)
        experts = [cls._extend_int_list(v, n_groups) for v in per_layer]
        dims = [cls._extend_int_list(v, 1) for v in list_of_dnn_dims]
        shared = [g < num_shared_task for g in range(n_groups)]
        # shared groups are named with range(num_task) in the reference (:108); identical whenever the constructor accepts
        names = ['shared_%d' % g for g in range(num_task)] + ['special_%d' % g for g in range(num_task)]
        return dims, experts, shared, names

    # -- build: modules + wiring plan -------------------------------------------------------------------------------
    def _expert_stack(self, layer_idx, group_name, widths, n_experts):
        scope = '%s/ple_layer_%s/task_%s' % (self.name, layer_idx, group_name)
        stack = [MultiDenseLayer(width, n_experts, activation=self.activation if i + 1 < len(widths) else None,
                                 name='%s/MultiDenseLayer_%d' % (scope, i), **self._dense_kwargs())
                 for i, width in enumerate(widths)]
        return _ExpertDNN(stack, scope)

    def build(self, input_shape):
        n_layers = len(self.list_of_dnn_dims)
        groups = list(zip(self.is_shared_tasks, self.task_names))           # zip truncation == the reference's
        shared_ids = [g for g, (is_shared, _) in enumerate(groups) if is_shared]
        self.dnns, self.gates, self._plan = [], [], []
        for l in range(n_layers):
            n_exp = self.list_of_num_experts_per_task[l]
            n_all = sum(n for n, _ in zip(n_exp, groups))
            n_shared = sum(n_exp[g] for g in shared_ids)
            stacks, gates, wiring = [], [], []
            for g, (is_shared, gname) in enumerate(groups):
                stacks.append(self._expert_stack(l, gname, self.list_of_dnn_dims[l], n_exp[g]))
                if is_shared and l == n_layers - 1:
                    gates.append(None)                                        # shared groups feed nobody after the last layer
                else:
                    units = n_all if is_shared else n_exp[g] + n_shared
                    gates.append(_GateDense(units, '%s/ple_gate_%s/task_%s/dense' % (self.name, l, gname)))
                # what this group reads / mixes: every group (shared) or itself followed by the shared groups (specific)
                sees = list(range(len(groups))) if is_shared else [g] + shared_ids
                wiring.append(sees)
            self.dnns.append(stacks)
            self.gates.append(gates)
            self._plan.append(wiring)
        # nn.Module bookkeeping (parameters(), .to(), state_dict())
        self._dnn_modules = nn.ModuleList([m for stacks in self.dnns for m in stacks])
        self._gate_modules = nn.ModuleList([m for gates in self.gates for m in gates if m is not None])
        self.built = True

    # -- call ---------------------------------------------------------------------------------------------------------
    def call(self, inputs):
        """inputs (B, D).  Returns the num_task task outputs (shared groups excluded), each (B, dim_out)."""
        prev = None                                             # gated outputs of the previous PLE layer, one per group
        for stacks, gates, wiring in zip(self.dnns, self.gates, self._plan):
            feeds = [inputs if prev is None else torch.cat([prev[s] for s in sees], dim=-1) for sees in wiring]
            expert_out = [stack(feed) for stack, feed in zip(stacks, feeds)]                  # per group: (N_g, B, dim_out)
            mixed = []
            for gate, feed, sees in zip(gates, feeds, wiring):
                if gate is None:
                    mixed.append(None)
                    continue
                visible = [e for s in sees for e in expert_out[s].unbind(0)]                  # no concat: a list of (B, dim_out)
                mixed.append(moe_mix(gate(feed), visible))                                    # softmax gate fused in the mix kernel
            prev = mixed
        return [out for out in prev if out is not None]
