"""gpyrn_amd -- MI355X-native mean-field inference for GP regression networks.

Drop-in for the hot path of iastro-pt/gpyrn (``import gpyrn_amd as gpyrn``):
same names as the reference's ``gpyrn/__init__.py:3-9``.  Importing the
package never touches the GPU; the HIP library (``libgprn_hip.so``) is loaded
on the first ELBO evaluation and its absence is an error, not a fallback.
"""
__version__ = '1.0'

from .meanfunc import Constant, Linear  # noqa: F401
from .covfunc import SquaredExponential, QuasiPeriodic  # noqa: F401
from .meanfield import inference  # noqa: F401
