"""TEST INFRASTRUCTURE ONLY: ctypes loader of oracle/libpairs_oracle.so (plain-C restatement, see pairs_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, 'libpairs_oracle.so')
        if not os.path.exists(so):
            subprocess.run(['make', '-C', _HERE, '-s'], check=True)
        _lib = ctypes.CDLL(so)
        _lib.oracle_pair_indices.restype = ctypes.c_int64
        _lib.oracle_pairwise_bpr.restype = ctypes.c_int64
        _lib.oracle_pairwise_bpr_grouped.restype = ctypes.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def pair_indices(groups, labels, scores, mask=None, flags=1):
    g = np.ascontiguousarray(groups, np.float32).reshape(-1)
    y = np.ascontiguousarray(labels, np.float32).reshape(-1)
    s = np.ascontiguousarray(scores, np.float32).reshape(-1)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(-1)
    B = g.size
    lib = _load()
    n = lib.oracle_pair_indices(_p(g), _p(y), _p(s), _p(m), ctypes.c_int64(B), flags, None, None, ctypes.c_int64(0))
    pos, neg = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    lib.oracle_pair_indices(_p(g), _p(y), _p(s), _p(m), ctypes.c_int64(B), flags, _p(pos), _p(neg), ctypes.c_int64(n))
    return pos[:n], neg[:n]


def pairwise_bpr(groups, labels, scores, mask=None, flags=1, factor=1.0, power=0.0, grouped=False):
    g = np.ascontiguousarray(groups, np.float32).reshape(-1)
    y = np.ascontiguousarray(labels, np.float32).reshape(-1)
    s = np.ascontiguousarray(scores, np.float32).reshape(-1)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(-1)
    B = g.size
    loss = ctypes.c_double(0.0)
    d = np.zeros(max(B, 1), np.float64)
    fn = _load().oracle_pairwise_bpr_grouped if grouped else _load().oracle_pairwise_bpr
    P = fn(_p(g), _p(y), _p(s), _p(m), ctypes.c_int64(B), flags, ctypes.c_double(factor),
                                    ctypes.c_double(power), ctypes.byref(loss), _p(d))
    return loss.value, d[:B], P
