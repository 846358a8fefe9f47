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


def parse(a):          # a BASELINE config number, or N,p,q,kind
    return synth.CONFIGS[int(a)] if a.isdigit() else (int(a.split(',')[0]), int(a.split(',')[1]), int(a.split(',')[2]), a.split(',')[3])


for shape in ([parse(a) for a in sys.argv[1:]] or [synth.CONFIGS[3], synth.CONFIGS[2], (512, 3, 2, 'QP')]):
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
            # ... and a whole evaluation (set-up + the warm-started loop), as bench.py --latency times it
            x0 = np.array(g.get_parameters(), dtype=float)
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                g.nELBO(x0 * 1.001)
                te = []
                for k in range(20):
                    t0 = time.perf_counter()
                    g.nELBO(x0 * (1.0 + 0.001 * (k % 5)))
                    te.append((time.perf_counter() - t0) * 1e3)
            print('N %d p %d q %d  set-up with %-12s: median %.3f ms  (min %.3f, max %.3f) | nELBO with new parameters: median %.3f ms' % (
                shape[0], shape[1], shape[2], name, np.median(ts), min(ts), max(ts), np.median(te)), flush=True)
    assert ctx.option('fallbacks') == 0
