"""Small host helpers shared by covfunc / meanfunc / meanfield.

Mirrors the only piece of the reference's `_utils.py` the hot path touches:
the `_array_input` decorator (gpyrn/_utils.py:20-27) and the `Array` alias
(:17).  The astro/statistics helpers in that file are out of scope
(SURVEY.md §2 row 6).
"""
from functools import wraps

import numpy as np

Array = np.ndarray


def _array_input(method):
    """Hand the wrapped method its argument as an (at least) 1-d ndarray."""
    @wraps(method)
    def with_array(self, t):
        return method(self, np.atleast_1d(t))
    return with_array


def _take_leading(owner, p, kind):
    """Parameter chaining shared by kernels and means (covfunc.py:30-41,
    meanfunc.py:23-34): consume the leading ``owner.pars.size`` entries of
    ``p``; return the remainder, or None when nothing is left over."""
    n = owner.pars.size
    assert len(p) >= n, f'too few parameters for {kind} {type(owner).__name__}'
    if len(p) == n:
        owner.pars = p
        return None
    owner.pars = np.array(p[:n], dtype=float)
    return np.array(p[n:])
