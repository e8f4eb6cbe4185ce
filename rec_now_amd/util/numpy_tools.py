"""Parity metric helpers -- same API as rec_now/util/numpy_tools.py (/root/reference/rec_now/util/numpy_tools.py:12-42).
torch tensors (CPU or GPU) are accepted wherever the reference accepted tf tensors."""
import numpy as np


def _to_numpy(a):
    if hasattr(a, 'detach'):
        a = a.detach().cpu().numpy()
    return a


def calc_sum_of_abs_diff(arr1, arr2):
    """Sum of |arr1 - arr2| in float64 (the tolerance metric of every reference test)."""
    arr1 = np.array(_to_numpy(arr1), dtype=np.float64)
    arr2 = np.array(_to_numpy(arr2), dtype=np.float64)
    return np.sum(np.abs(arr1 - arr2))


def all_equal(arr1, arr2):
    """True when every element of arr1 equals the matching element of arr2."""
    return bool(np.all(np.array(_to_numpy(arr1)) == np.array(_to_numpy(arr2))))
