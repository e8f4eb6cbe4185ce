"""Minimal Keras-style layer plumbing on torch.nn.Module, so the reference's Layer API carries over:
`Layer(**ctor)(inputs, ...)` -> lazy `build(input_shape)` on first call -> `call(inputs, ...)`; weights are created with
`add_weight(name, shape, initializer)` and keep the reference's variable names (exposed through `named_weights()` /
`state_dict()`), so reference-named checkpoints load 1:1.

Only what the hot-path layers use from `keras.layers.Layer` / `keras.layers.Dense` is provided.
"""
import math

import torch
from torch import nn

ACT_LINEAR, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
_ACT_CODES = {None: ACT_LINEAR, 'linear': ACT_LINEAR, 'relu': ACT_RELU, 'tanh': ACT_TANH, 'sigmoid': ACT_SIGMOID}


def activation_code(activation):
    """keras activation identifier -> (fused kernel code or None, python callable or None).
    Names the kernels fuse: None/'linear', 'relu', 'tanh', 'sigmoid'.  Any other value must be a callable on torch
    tensors; it is applied unfused after the kernel."""
    if activation is None or isinstance(activation, str):
        if activation not in _ACT_CODES:
            raise ValueError('Unknown activation function: %s (fused: linear, relu, tanh, sigmoid; '
                             'pass a callable for anything else)' % activation)
        return _ACT_CODES[activation], None
    if callable(activation):
        return None, activation
    raise ValueError('Could not interpret activation function identifier: %r' % (activation,))


def _fans(shape):
    shape = tuple(int(s) for s in shape)
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape
    rfs = 1
    for s in shape[:-2]:
        rfs *= s
    return shape[-2] * rfs, shape[-1] * rfs


def get_initializer(identifier):
    """keras.initializers.get for the identifiers the hot path uses; callables `f(shape) -> tensor` pass through."""
    if callable(identifier):
        return identifier
    if identifier in (None, 'glorot_uniform'):
        def glorot_uniform(shape, generator=None):
            fi, fo = _fans(shape)
            limit = math.sqrt(3.0 / max(1.0, (fi + fo) / 2.0))
            return (torch.rand(tuple(shape), generator=generator) * 2.0 - 1.0) * limit
        return glorot_uniform
    if identifier == 'glorot_normal':
        def glorot_normal(shape, generator=None):
            fi, fo = _fans(shape)
            std = math.sqrt(1.0 / max(1.0, (fi + fo) / 2.0)) / 0.87962566103423978
            return torch.nn.init.trunc_normal_(torch.empty(tuple(shape)), 0.0, std, -2 * std, 2 * std, generator=generator)
        return glorot_normal
    if identifier == 'zeros':
        return lambda shape, generator=None: torch.zeros(tuple(shape))
    if identifier == 'ones':
        return lambda shape, generator=None: torch.ones(tuple(shape))
    if identifier == 'random_normal':
        return lambda shape, generator=None: torch.randn(tuple(shape), generator=generator) * 0.05
    if identifier == 'random_uniform':
        return lambda shape, generator=None: torch.rand(tuple(shape), generator=generator) * 0.1 - 0.05
    raise ValueError('Unknown initializer: %r' % (identifier,))


def _shape_of(x):
    if isinstance(x, (list, tuple)):
        return [_shape_of(v) for v in x]
    return tuple(x.shape)


_SAFE = str.maketrans({'/': '__', '.': '_'})


class Layer(nn.Module):
    """keras.layers.Layer look-alike: lazy build, add_weight, call."""

    def __init__(self, trainable=True, name=None, dtype=None, dynamic=False, **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError('unexpected keyword arguments: %s' % sorted(kwargs))
        self.trainable = trainable
        self.name = name if name is not None else type(self).__name__.lower()
        self.dtype = dtype or 'float32'
        self.built = False
        self._weight_names = {}
        self._build_device = None

    # -- keras API -------------------------------------------------------------------------------
    def add_weight(self, name, shape, initializer=None, regularizer=None, constraint=None, dtype=None, trainable=True):
        init = get_initializer(initializer)
        value = init(tuple(int(s) for s in shape)).to(torch.float32)
        param = nn.Parameter(value.to(self._build_device or 'cpu'), requires_grad=bool(trainable and self.trainable))
        attr = 'w_' + name.translate(_SAFE)
        self.register_parameter(attr, param)
        self._weight_names[name] = attr
        param.regularizer = regularizer
        param.constraint = constraint
        return param

    def build(self, input_shape):
        self.built = True

    def call(self, inputs, *args, **kwargs):
        raise NotImplementedError

    def forward(self, inputs, *args, **kwargs):
        if not self.built:
            first = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
            self._build_device = first.device if isinstance(first, torch.Tensor) else None
            self.build(_shape_of(inputs))
            self.built = True
        return self.call(inputs, *args, **kwargs)

    # -- weights by their reference (Keras) names ----------------------------------------------------
    def named_weights(self, prefix=''):
        out = {}
        for name, attr in self._weight_names.items():
            out[prefix + name] = getattr(self, attr)
        def walk(module, pre):
            for _, child in module.named_children():
                if isinstance(child, Layer):
                    out.update(child.named_weights(pre + child.name + '/'))
                else:                      # nn.ModuleList and other plain containers are transparent
                    walk(child, pre)
        walk(self, prefix)
        return out

    def set_weights_by_name(self, values):
        """values: dict reference-variable-name -> array-like.  The layer must be built."""
        weights = self.named_weights()
        with torch.no_grad():
            for k, v in values.items():
                if k not in weights:
                    raise KeyError('%s has no weight named %r (has: %s)' % (self.name, k, sorted(weights)))
                w = weights[k]
                w.copy_(torch.as_tensor(v, dtype=w.dtype).reshape(w.shape))

    @property
    def losses(self):
        """Regularization terms, as keras collects them from add_weight(regularizer=...)."""
        out = []
        for p in self.parameters():
            reg = getattr(p, 'regularizer', None)
            if reg is not None:
                out.append(reg(p))
        return out


class DenseBase(Layer):
    """The constructor contract of keras.layers.Dense that DCNLayer / MultiDenseLayer / MMOELayer / PLELayer inherit
    (reference: `class DCNLayer(keras.layers.Dense)` and friends call `super().__init__(units, **kwargs)`)."""

    def __init__(self, units, activation=None, use_bias=True, kernel_initializer='glorot_uniform',
                 bias_initializer='zeros', kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 kernel_constraint=None, bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.units = int(units) if not isinstance(units, int) else units
        if self.units < 0:
            raise ValueError('Received an invalid value for `units`, expected a positive integer, got %s.' % units)
        self.activation = activation
        self.act_code, self.act_callable = activation_code(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer
        self.activity_regularizer = activity_regularizer
        self.kernel_constraint = kernel_constraint
        self.bias_constraint = bias_constraint

    def _dense_kwargs(self):
        return dict(use_bias=self.use_bias, kernel_constraint=self.kernel_constraint,
                    bias_constraint=self.bias_constraint, kernel_regularizer=self.kernel_regularizer,
                    bias_regularizer=self.bias_regularizer, activity_regularizer=self.activity_regularizer,
                    kernel_initializer=self.kernel_initializer, bias_initializer=self.bias_initializer)
