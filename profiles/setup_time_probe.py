"""Round 6: what the substitution panels cost the set-up (gprn_factor_priors: fills, chol(K), chol(K)^-1, K_j^-1).
    python profiles/setup_time_probe.py [config ...]          (on the GPU box; default configs 3 2 and N = 512, p = 3, q = 2)
accurate_factor 0 = panel steps as products with explicit inverses (rounds 1-5), default = substitution (round 6)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import gpyrn_amd as gpyrn   # noqa: E402
from gpyrn_amd import covfunc, meanfunc, synth   # noqa: E402


def problem(N, p, q, kind):
    t, ys, es = synth.rv_series(N, p)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, synth.component_spec(p, q, kind))
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    return g


for shape in ([synth.CONFIGS[int(a)] for a in sys.argv[1:]] or [synth.CONFIGS[3], synth.CONFIGS[2], (512, 3, 2, 'QP')]):
    g = problem(*shape)
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    for rep in range(2):
        for acc, name in ((0, 'products'), (-2, 'substitution')):
            ctx.option('accurate_factor', acc)
            ctx.factor_priors()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                info = ctx.factor_priors()
                ts.append((time.perf_counter() - t0) * 1e3)
            assert info == 0
            print('N %d p %d q %d  set-up with %-12s: median %.3f ms  (min %.3f, max %.3f)' % (
                shape[0], shape[1], shape[2], name, np.median(ts), min(ts), max(ts)), flush=True)
    assert ctx.option('fallbacks') == 0
