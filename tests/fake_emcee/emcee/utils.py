"""Stand-in for emcee.utils (TEST INFRASTRUCTURE)."""
import numpy as np


def sample_ellipsoid(p0, covmat, size=1):
    """`size` draws from N(p0, covmat), from NumPy's global generator (emcee.utils.sample_ellipsoid)."""
    return np.random.multivariate_normal(np.atleast_1d(p0), np.atleast_2d(covmat), size=size)
