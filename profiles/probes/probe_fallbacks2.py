"""Which factor_priors call of a small model hits the wait budget on the flag schedule?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc

which = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
for N in (300, 200, 384, 700):
    rng = np.random.default_rng(5)
    p, q = 2, 3
    t = np.sort(rng.uniform(0, 60, N))
    args = []
    for _ in range(p):
        args += [rng.normal(size=N), rng.uniform(0.1, 0.3, N)]
    g = gpyrn.inference(q, t, *args)
    if which == 'mixed':
        nodes = [covfunc.SquaredExponential(1.0, 4.0), covfunc.Periodic(1.0, 11.0, 0.8),
                 covfunc.QuasiPeriodic(1.0, 20.0, 9.0, 0.7)]
    else:
        nodes = [covfunc.SquaredExponential(1.0, 4.0), covfunc.SquaredExponential(1.0, 5.0),
                 covfunc.SquaredExponential(1.0, 6.0)]
    weights = [covfunc.SquaredExponential(0.8, 15.0), covfunc.Matern32(0.9, 12.0)] * q
    g.set_components(nodes, weights, [None] * p, [0.2] * p)
    for rep in range(3):
        t0 = time.time()
        E = g.ELBOcalc(max_iter=5)
        print(f'{which} N={N} rep={rep}: ELBO {E[0]:.6f} {1e3 * (time.time() - t0):.0f} ms, info {g.last_info}', flush=True)
