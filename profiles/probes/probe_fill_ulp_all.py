"""Every built-in kernel's device fill against NumPy's evaluation of the reference's formula (gpyrn_amd.covfunc.__call__,
pinned to the reference's matrices by tests/golden/kernels.npz) at N = 1000 over 800 days: ulp statistics.
    python profiles/probes/probe_fill_ulp_all.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, '.')
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc
from tests import _cases
meta = json.load(open(os.path.join(_cases.GOLDEN, 'kernels.json')))
rng = np.random.RandomState(3)
n = 1000
t = np.sort(rng.uniform(0.0, 800.0, n))
g = gpyrn.inference(1, t, np.zeros(n), np.ones(n))
r = t[:, None] - t[None, :]
cases = [(name, getattr(covfunc, name)(*pars)) for name, pars in meta['simple']] + [(tag, eval(expr, {'c': covfunc})) for tag, expr in meta['composite']]
for name, k in cases:
    if k._device_program() is None:
        print('%-24s no device program' % name); continue
    K = g._KMatrix(k)
    two_arg = isinstance(k, tuple(getattr(covfunc, x) for x in ('Paciorek',) if hasattr(covfunc, x)))
    try:
        want = k(r) + (1e-6 * np.eye(n))
    except Exception as e:
        print('%-24s host evaluation failed: %s' % (name, e)); continue
    if want.shape != K.shape: print(name, 'shape'); continue
    scale = np.abs(want).max()
    rel = np.abs(want) > 1e-6 * scale
    ulp = np.abs(K - want)[rel] / np.spacing(np.abs(want[rel]))
    print('%-24s mean %8.2f ulp  99.9%% %8.0f  max %10.0f   max |dK| / max |K| %.1e' % (name, ulp.mean(), np.percentile(ulp, 99.9), ulp.max(), np.abs(K - want).max() / scale), flush=True)
