"""Side-by-side evaluations for a profiler: `python profiles/batch_run.py N p q B [calls]` runs inference.nELBO_batch on the
synthetic problem of bench.py --latency (perturbed hyper-parameters, warm start) `calls` times after one warm-up call and
prints evaluations/s of the best call.  Used under rocprofv3 --kernel-trace --stats (profiles/r05_batch_*)."""
import contextlib
import io
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import gpyrn_amd as gpyrn                                    # noqa: E402
from gpyrn_amd import covfunc, meanfunc, synth               # noqa: E402

N, p, q, B = (int(v) for v in sys.argv[1:5])
calls = int(sys.argv[5]) if len(sys.argv) > 5 else 3
kind = 'SE' if (p, q) == (1, 1) and N <= 200 else 'QP'
t, ys, es = synth.rv_series(N, p)
spec = synth.component_spec(p, q, kind)
nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair])
g.set_components(nodes, weights, means, jit)
x0 = np.array(g.get_parameters(), dtype=float)
rng = np.random.RandomState(1)
xb = [x0 * (1.0 + 0.01 * rng.standard_normal(x0.size)) for _ in range(B)]
best = None
ctx = g._backend()
lib_s = []
inner = ctx.elbocalc_batch


def timed(*a, **k):                                        # the library's share of a call (ctypes included)
    t0 = time.perf_counter()
    out = inner(*a, **k)
    lib_s.append(time.perf_counter() - t0)
    return out


ctx.elbocalc_batch = timed
with contextlib.redirect_stdout(io.StringIO()):
    g.nELBO(x0)
    g.nELBO_batch(xb)
    for _ in range(calls):
        t0 = time.perf_counter()
        g.nELBO_batch(xb)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, best_lib = dt, lib_s[-1]
print('python side of the best call: %.0f us around %.0f us in gprn_elbocalc_batch' % (1e6 * (best - best_lib), 1e6 * best_lib))
print('N=%d p=%d q=%d B=%d: %.3f ms per call, %.0f evaluations/s; flags %d fallbacks %d' % (
    N, p, q, B, 1e3 * best, B / best, ctx.option('flags'), ctx.option('fallbacks')))
