"""Synthetic radial-velocity series used by bench.py and the parity tests.

This is the generator SURVEY.md §8(d) / BASELINE.md §3 specify (it is not part
of the reference): legacy ``RandomState`` so the stream is stable across NumPy
versions.  Draw order matters and is: t, then per output (yerr_i, noise_i).
"""
import numpy as np

# (N, p, q, node kernel) of BASELINE.json `configs`, 1-based
CONFIGS = {
    1: (200, 1, 1, 'SE'),
    2: (2048, 1, 1, 'QP'),
    3: (4096, 3, 2, 'QP'),
    4: (4096, 3, 4, 'QP'),
    5: (16384, 4, 3, 'QP'),
}


def rv_series(N, p, seed=0):
    """Return (t, [y_0..y_{p-1}], [yerr_0..yerr_{p-1}])."""
    rng = np.random.RandomState(seed)
    t = np.sort(rng.uniform(0.0, 0.4 * N, N))
    ys, es = [], []
    for i in range(p):
        yerr = rng.uniform(0.5, 1.5, N)
        y = (5 + i) * np.sin(2 * np.pi * t / 25 + 0.3 * i) * (1 + 0.002 * t) \
            + rng.normal(0.0, yerr)
        ys.append(y)
        es.append(yerr)
    return t, ys, es


def component_spec(p, q, node_kind='QP'):
    """Hyper-parameters of the benchmark model as plain (name, params) tuples."""
    nodes = []
    for j in range(q):
        if node_kind == 'SE':
            nodes.append(('SquaredExponential', [1 + 0.1 * j, 20.0 + j]))
        else:
            nodes.append(('QuasiPeriodic', [1 + 0.1 * j, 50.0 + j, 25.0, 0.7]))
    weights = [('SquaredExponential', [1 + 0.05 * k, 60.0 + k])
               for k in range(q * p)]
    means = [('Constant', [0.0]) for _ in range(p)]
    jitters = [0.5] * p
    return nodes, weights, means, jitters


def build_components(covfunc, meanfunc, spec):
    """Instantiate a spec against a covfunc/meanfunc module pair."""
    nodes, weights, means, jitters = spec

    def mk(mod, item):           # (name, [parameters]); a parameter that is itself such a pair is built first (Sum, Multiplication)
        if item is None:
            return None
        return getattr(mod, item[0])(*[mk(mod, a) if isinstance(a, (list, tuple)) and a and isinstance(a[0], str) else a
                                       for a in item[1]])
    return ([mk(covfunc, n) for n in nodes], [mk(covfunc, w) for w in weights],
            [mk(meanfunc, m) for m in means], list(jitters))
