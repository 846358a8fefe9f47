"""Dataflow schedule (queue.hip) on caller matrices: right factors, no fallback, time per factorisation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from gpyrn_amd import _hip

def spd(n, rng, shift=1.0):
    t = np.sort(rng.uniform(0, 0.4 * n, n))
    r = t[:, None] - t[None, :]
    return np.exp(-0.5 * r**2 / 30.0**2) + shift * np.eye(n)

ctx = _hip.Context(0)
ctx.option('wait_budget_ms', int(os.environ.get('BUDGET_MS', 500)))
print('flags', ctx.option('flags'), 'queue', ctx.option('queue'), flush=True)
rng = np.random.RandomState(3)
for n, batch in [(256, 1), (1024, 1), (1024, 3), (2048, 2), (4096, 2), (4096, 6)]:
    A = np.array([spd(n, rng, 1.0 + 0.3 * b) for b in range(batch)])
    t0 = time.time()
    L, X, info = ctx.test_factor_invert(A)
    dt = time.time() - t0
    errL = max(np.abs(np.tril(L[b]) - np.linalg.cholesky(A[b])).max() for b in range(batch))
    errX = max(np.abs(np.tril(X[b]) @ np.tril(L[b]) - np.eye(n)).max() for b in range(batch))
    print(f'n={n} batch={batch} info={info} errL={errL:.2e} errX={errX:.2e} fallbacks={ctx.option("fallbacks")} '
          f'queue={ctx.option("queue")} flags={ctx.option("flags")} wall={dt:.3f}s', flush=True)
    if ctx.option('fallbacks'):
        break
