"""How many live contexts (4 streams each) until the flag schedule's in-kernel waits start timing out?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from gpyrn_amd import _hip

rng = np.random.RandomState(3)
n, batch = 384, 9
t = np.sort(rng.uniform(0, 50, n))
A = np.array([(1.0 + b) * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 4.0) + np.eye(n) for b in range(batch)])
live = []
for k in range(1, 17):
    c = _hip.Context(0)
    live.append(c)
    c.option('wait_budget_ms', 200)
    for use_all in (0, 1):
        tot = 0
        t0 = time.time()
        for cc in (live if use_all else [c]):
            f0 = cc.option('fallbacks')
            for rep in range(3):
                L, X, info = cc.test_factor_invert(A)
                cc.option('flags', 1)
            tot += cc.option('fallbacks') - f0
        print(f'live={k} {"every context" if use_all else "newest only"}: fallbacks={tot} {1e3 * (time.time() - t0):.0f} ms', flush=True)
