"""A small, deterministic stand-in for the `emcee` package.  TEST INFRASTRUCTURE.

emcee is not installed in this image, so `inference.mcmc` could never run.  This module implements
just the surface `mcmc` touches (meanfield.py:1154-1286) -- EnsembleSampler with the Goodman & Weare
stretch move, `sample()` as a generator, `iteration`, `get_autocorr_time(tol=0)`, `get_chain`,
`get_log_prob`, `get_blobs`; `backends.HDFBackend` (in memory) and `utils.sample_ellipsoid` -- with
every random number drawn from NumPy's global generator in a fixed order.  The SAME module is put in
front of the reference (oracle/gen_golden.py) and of gpyrn_amd (tests), so the two chains can be compared
number by number.  It makes no claim to be a good sampler.
"""
import numpy as np

from . import backends, utils  # noqa: F401

__version__ = '0.0-fake'


class _State:
    def __init__(self, coords, log_prob, blobs):
        self.coords, self.log_prob, self.blobs = coords, log_prob, blobs


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, backend=None, a=2.0, vectorize=False, **kwargs):
        self.nwalkers, self.ndim, self.fn, self.a = int(nwalkers), int(ndim), log_prob_fn, float(a)
        self.pool, self.backend, self.vectorize = pool, backend, bool(vectorize)
        self.iteration = 0
        self._chain, self._lp, self._blobs = [], [], []

    def _evaluate(self, points):
        if self.vectorize:                       # emcee: the function gets all coordinates at once, one row per walker back
            res = [tuple(row) for row in np.atleast_2d(self.fn(np.array(points)))]
        else:
            mapper = self.pool.map if self.pool is not None else map
            res = list(mapper(self.fn, [np.array(p) for p in points]))
        lp = np.array([r[0] if isinstance(r, tuple) else r for r in res], dtype=float)
        bl = np.array([r[1] if isinstance(r, tuple) else np.nan for r in res], dtype=float)
        return lp, bl

    def sample(self, initial_state, iterations=1, progress=False, **kwargs):
        x = np.array(initial_state, dtype=float)
        lp, bl = self._evaluate(x)
        half = self.nwalkers // 2
        for _ in range(int(iterations)):
            for first, other in ((slice(0, half), slice(half, None)), (slice(half, None), slice(0, half))):
                s, c = x[first], x[other]
                ns = s.shape[0]
                z = ((self.a - 1.0) * np.random.rand(ns) + 1.0) ** 2 / self.a
                partner = np.random.randint(c.shape[0], size=ns)
                q = c[partner] + z[:, None] * (s - c[partner])
                lq, bq = self._evaluate(q)
                lnratio = (self.ndim - 1.0) * np.log(z) + lq - lp[first]
                accept = np.log(np.random.rand(ns)) < lnratio
                idx = np.arange(self.nwalkers)[first][accept]
                x[idx], lp[idx], bl[idx] = q[accept], lq[accept], bq[accept]
            self.iteration += 1
            self._chain.append(x.copy()); self._lp.append(lp.copy()); self._blobs.append(bl.copy())
            yield _State(x.copy(), lp.copy(), bl.copy())

    def get_chain(self, flat=False, **kwargs):
        c = np.array(self._chain)
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, flat=False, **kwargs):
        c = np.array(self._lp)
        return c.ravel() if flat else c

    def get_blobs(self, flat=False, **kwargs):
        c = np.array(self._blobs)
        return c.ravel() if flat else c

    def get_autocorr_time(self, tol=0, **kwargs):
        """Integrated autocorrelation time of the ensemble mean, per dimension (window = n/2)."""
        c = np.array(self._chain).mean(axis=1)                      # (n, ndim)
        n = c.shape[0]
        tau = np.ones(self.ndim)
        for d in range(self.ndim):
            y = c[:, d] - c[:, d].mean()
            v = float(np.dot(y, y))
            if n < 4 or v <= 0.0:
                continue
            rho = np.array([np.dot(y[:n - k], y[k:]) / v for k in range(1, n // 2)])
            tau[d] = 1.0 + 2.0 * float(np.sum(rho))
        return np.abs(tau) + 1e-12
