"""Rebuild model problems from the committed golden fixtures (data only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(tag):
    with open(os.path.join(GOLDEN, tag + '.json')) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, tag + '.npz'))
    return meta, data


def available(tag):
    return os.path.exists(os.path.join(GOLDEN, tag + '.npz'))


def components(meta, covfunc, meanfunc):
    def mk(mod, item):           # (composite kernels: a parameter that is itself (name, [parameters]) is built first)
        if item is None:
            return None
        return getattr(mod, item[0])(*[mk(mod, a) if isinstance(a, (list, tuple)) and a and isinstance(a[0], str) else a
                                       for a in item[1]])
    nodes = [mk(covfunc, n) for n in meta['nodes']]
    weights = [mk(covfunc, w) for w in meta['weights']]
    means = [mk(meanfunc, m) for m in meta['means']]
    return nodes, weights, means, list(meta['jitters'])


def data_args(data):
    args = []
    for y, e in zip(data['y'], data['yerr']):
        args += [np.array(y), np.array(e)]
    return args


# ---- north_star's tolerance on the posterior state: "within 1e-8 relative on fp64 for the ELBO and posterior means"
# (/root/reference/gpyrn/meanfield.py:792, 865 produce the means; :688-697 the variances).  Relative to what: the state of
# a latent GP is a vector of N values that pass through zero, so the bound is norm-wise PER LATENT GP -- the largest
# deviation of a row of the (p+1, q, N) state against that row's largest entry.  Every check is recorded, and a session
# that ran any writes the achieved figures to gpurun_out/parity_achieved.txt (tests/conftest.py).
STATE_TOL = 1e-8
ACHIEVED = []


def _rowwise(a, ref):
    a = np.asarray(a, dtype=float)
    ref = np.asarray(ref, dtype=float)
    assert a.size == ref.size, (a.shape, ref.shape)
    ref = ref.reshape(a.shape)
    a2 = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(1, -1)
    r2 = ref.reshape(a2.shape)
    scale = np.abs(r2).max(axis=1)
    scale = np.where(scale > 0, scale, 1.0)
    return float((np.abs(a2 - r2).max(axis=1) / scale).max())


def assert_state(what, mu, mu_ref, var=None, var_ref=None, tol=STATE_TOL):
    """max |mu - mu_ref| <= tol max |mu_ref| per latent GP (row of the state), the same for the variances."""
    e_mu = _rowwise(mu, mu_ref)
    e_var = _rowwise(var, var_ref) if var is not None else None
    ACHIEVED.append((what, e_mu, e_var))
    assert e_mu <= tol, '%s: posterior means off by %.3g (norm-wise per latent GP; bound %.1g)' % (what, e_mu, tol)
    if e_var is not None:
        assert e_var <= tol, '%s: posterior variances off by %.3g (norm-wise per latent GP; bound %.1g)' % (what, e_var, tol)
