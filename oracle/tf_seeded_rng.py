"""TEST INFRASTRUCTURE ONLY (oracle/): never imported by the product package.

Restatement of TensorFlow's seeded stateful random ops (TF <= 2.6 semantics), so that the
reference's unit-test inputs/weights can be regenerated WITHOUT TensorFlow (TF is not
installable in this image).  The reference goldens are functions of this RNG:

  * /root/reference/tests/layers/test_fm_layer.py:19-27        tf.random.set_seed(1) + tf.random.uniform(seed=f)
  * /root/reference/tests/layers/test_dcn_layer.py:19-22       glorot_uniform kernels (keras Dense default)
  * /root/reference/tests/layers/test_multi_dense_layer.py:22-32
  * /root/reference/tests/layers/test_mmoe_layer.py:19-26
  * /root/reference/tests/layers/test_dcn_mix_layer.py:21-24
  * /root/reference/tests/layers/test_cin_layer.py:19-34
  * /root/reference/tests/layers/test_ple_layer.py:19-26

Published algorithm restated here (third-party dependency of the reference: TensorFlow 2.x,
version unpinned -- the reference has no requirements file):

  1. tf.random.set_seed(g): global seed g, and a CPython random.Random(g); every random op
     created WITHOUT an op seed draws op_seed = rng.randint(0, 2**31 - 1) in program order.
     Ops with seed=s use op_seed = s and do not consume from that stream.
  2. Each op instance = fresh PhiloxRandom(key=g, counter[2:4]=op_seed), Philox4x32-10.
  3. uint32 -> float in [0,1): bitcast((127 << 23) | (x & 0x7fffff)) - 1.0f.
  4. random_uniform(a, b) = u*(b-a)+a in fp32.  glorot_uniform limit = sqrt(3/max(1,(fi+fo)/2)).
  5. random_normal: Box-Muller per pair of uint32, u1 = max(f(x0),1e-7), v1 = 2*pi*f(x1),
     r = sqrt(-2 ln u1), outputs sin(v1)*r, cos(v1)*r.
"""
import math
import random

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def _philox4x32_10(counters, key):
    """counters: (n,4) uint32 array; key: (k0,k1) python ints. Returns (n,4) uint32."""
    c = counters.astype(np.uint64)
    c0, c1, c2, c3 = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    k0, k1 = key
    for _ in range(10):
        p0 = _M0 * c0            # 64-bit products of 32-bit values
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def _philox_uint32(global_seed, op_seed, n):
    """First n uint32 outputs of PhiloxRandom(global_seed, op_seed)."""
    ngroups = (n + 3) // 4
    ctr = np.zeros((ngroups, 4), dtype=np.uint32)
    idx = np.arange(ngroups, dtype=np.uint64)
    ctr[:, 0] = (idx & _MASK32).astype(np.uint32)
    ctr[:, 1] = (idx >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = op_seed & 0xFFFFFFFF
    ctr[:, 3] = (op_seed >> 32) & 0xFFFFFFFF
    key = (global_seed & 0xFFFFFFFF, (global_seed >> 32) & 0xFFFFFFFF)
    return _philox4x32_10(ctr, key).reshape(-1)[:n]


def _uint32_to_float(x):
    bits = (np.uint32(127) << np.uint32(23)) | (x & np.uint32(0x7FFFFF))
    return bits.view(np.float32) - np.float32(1.0)


class TFSeededRNG:
    """State of `tf.random.set_seed(g)` in eager mode."""

    def __init__(self, global_seed):
        self.g = int(global_seed)
        self._rng = random.Random(self.g)

    def _op_seed(self, seed=None):
        if seed is not None:
            return int(seed)
        return self._rng.randint(0, 2 ** 31 - 1)

    def uniform(self, shape, minval=0.0, maxval=1.0, seed=None):
        n = int(np.prod(shape))
        u = _uint32_to_float(_philox_uint32(self.g, self._op_seed(seed), n))
        out = u * np.float32(maxval - minval) + np.float32(minval)
        return out.astype(np.float32).reshape(shape)

    def normal(self, shape, mean=0.0, stddev=1.0, seed=None):
        n = int(np.prod(shape))
        npair = (n + 1) // 2
        x = _philox_uint32(self.g, self._op_seed(seed), 2 * npair)
        u1 = np.maximum(_uint32_to_float(x[0::2]), np.float32(1.0e-7))
        v1 = np.float32(2.0 * math.pi) * _uint32_to_float(x[1::2])
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        out = np.empty(2 * npair, dtype=np.float32)
        out[0::2] = np.sin(v1).astype(np.float32) * r
        out[1::2] = np.cos(v1).astype(np.float32) * r
        out = out[:n] * np.float32(stddev) + np.float32(mean)
        return out.astype(np.float32).reshape(shape)

    # keras initializers with seed=None ------------------------------------------------
    def glorot_uniform(self, shape):
        shape = tuple(int(s) for s in shape)
        if len(shape) < 1:
            fan_in = fan_out = 1
        elif len(shape) == 1:
            fan_in = fan_out = shape[0]
        elif len(shape) == 2:
            fan_in, fan_out = shape
        else:
            rfs = int(np.prod(shape[:-2]))
            fan_in, fan_out = shape[-2] * rfs, shape[-1] * rfs
        limit = math.sqrt(3.0 / max(1.0, (fan_in + fan_out) / 2.0))
        return self.uniform(shape, -limit, limit)

    def random_normal_initializer(self, shape):
        """tf.random_normal_initializer() defaults: mean 0, stddev 0.05."""
        return self.normal(shape, 0.0, 0.05)
