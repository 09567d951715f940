"""Small helpers the reference exports at package level (reference mbb_emcee/utility.py)."""
import numpy as np

__all__ = ["isiterable"]


def isiterable(obj):
    """Can `obj` be looped over?  A 0-d numpy array cannot (it is an ndarray, and iterating it raises): utility.py:4-21."""
    if isinstance(obj, np.ndarray):
        return obj.ndim > 0
    try:
        iter(obj)
    except TypeError:
        return False
    return True
