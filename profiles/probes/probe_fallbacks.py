"""Do small / odd tile counts ever hit the wait budget on the flag schedule?  Prints the fallback count per shape."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from gpyrn_amd import _hip

rng = np.random.RandomState(3)


def spd(n, s):
    t = np.sort(rng.uniform(0, 50, n))
    return s * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 4.0) + np.eye(n)


for fresh in (0, 1):
    c = _hip.Context(0)
    for n in (128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1408, 1664):
        for batch in (1, 9):
            if fresh:
                c.close(); c = _hip.Context(0)
            A = np.array([spd(n, 1.0 + b) for b in range(batch)])
            f0 = c.option('fallbacks')
            t0 = time.time()
            for rep in range(3):
                L, X, info = c.test_factor_invert(A)
                c.option('flags', 1)
            dt = time.time() - t0
            err = max(np.abs(np.tril(L[b]) - np.linalg.cholesky(A[b])).max() for b in range(batch))
            print(f'fresh={fresh} n={n} T={(n + 127) // 128} batch={batch} fallbacks={c.option("fallbacks") - f0} '
                  f'info={info} err={err:.1e} {dt * 1e3:.0f} ms', flush=True)
    c.close()
