"""CINLayer -- drop-in for rec_now/layers/cin_layer.py (/root/reference/rec_now/layers/cin_layer.py:12-122):
the Compressed Interaction Network of xDeepFM."""
import torch

from ._keras import Layer
from ._ops import CINFunction


class CINLayer(Layer):
    """Symbols: B batch size, D embedding dim, F number of fields, Hs hidden channel sizes (Hs[0] = F)."""

    def __init__(self, hidden_sizes, embedding_dim=-1, initializer='glorot_uniform',
                 trainable=True, name=None, dtype=None, dynamic=False, **kwargs):
        """hidden_sizes: channels of every hidden layer; embedding_dim: D, required when `call` gets one tensor
        instead of a list of F embeddings."""
        super().__init__(trainable=trainable, name=name, dtype=dtype, dynamic=dynamic, **kwargs)
        self.hidden_sizes = list(hidden_sizes)
        self.embedding_dim = embedding_dim
        self.initializer = initializer

    def build(self, input_shape):
        """F and D from the input (a list of F (B, D) shapes, or one (B, F*D) shape with `embedding_dim` given); then one
        weight per hidden layer k >= 1: `weight_of_layer{k}` of shape [1, 1, H_k, H_{k-1} * F] with H_0 = F."""
        if isinstance(input_shape, list):
            self.num_field, self.embedding_dim = len(input_shape), int(input_shape[0][-1])
        elif self.embedding_dim > 0:
            self.num_field = int(int(input_shape[-1]) / self.embedding_dim)
        else:
            raise ValueError('embedding_dim shall bigger than 0 when inputs is not a list of embeddings.')
        widths = [self.num_field] + self.hidden_sizes
        self.idx2weight = {
            k: self.add_weight('weight_of_layer%s' % k, shape=[1, 1, widths[k], widths[k - 1] * self.num_field],
                               initializer=self.initializer, dtype=self.dtype, trainable=True)
            for k in range(1, len(widths))}
        self.built = True

    def call(self, inputs, output_input=True, sum_channel=True):
        """inputs: list of F tensors (B, D), or one (B, F*D) tensor.
        Returns (B, D) when sum_channel; else (B, sum(kept channels) * D) with kept = [F +] Hs[1:] (output_input
        keeps the F input fields; the reference docstring has the two cases swapped, the code is authoritative)."""
        if isinstance(inputs, (list, tuple)):
            emb = torch.cat(list(inputs), dim=1)      # (B, F*D)
        else:
            emb = inputs
        if emb.shape[-1] != self.num_field * self.embedding_dim:
            raise ValueError('expected %d x %d input features, got %d' % (self.num_field, self.embedding_dim, emb.shape[-1]))
        weights = [self.idx2weight[i] for i in range(1, len(self.hidden_sizes) + 1)]
        return CINFunction.apply(emb, self.embedding_dim, self.num_field, tuple(self.hidden_sizes), bool(output_input),
                                 bool(sum_channel), *weights)
