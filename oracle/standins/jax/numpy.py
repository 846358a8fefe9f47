"""jax.numpy stand-in: the NumPy namespace itself (fp64 by default)."""
from numpy import *  # noqa: F401,F403
from numpy import linalg, ndarray  # noqa: F401
