"""Test tooling only: NumPy/SciPy stand-in for the absent `jax` package.

Used solely by oracle/gen_golden.py so that /root/reference/gpyrn can be
imported (read-only) in the build container to generate golden vectors.
The reference uses jax only as a CPU JIT for fp64 LAPACK calls
(meanfield.py:6-9,71,895,992,1069), so NumPy's LAPACK is the same arithmetic.
Never shipped to / imported on the GPU box and never imported by gpyrn_amd.
"""
import functools as _ft


class _Config:
    def update(self, *a, **k):
        return None


config = _Config()


def jit(fun=None, **kwargs):
    # identity "JIT"; accepts jax.jit(f) and partial(jax.jit, static_argnums=..)(f)
    if fun is None:
        return lambda f: f
    return fun


from . import numpy  # noqa: E402,F401
from . import scipy  # noqa: E402,F401
