"""Round 6: the same check as factor_accuracy_probe.py on the launch schedule's THROUGHPUT task lists (outer panels of four
tiles, K = 512 updates: batch x tiles > 32) -- five copies of one ill-conditioned matrix at N = 1000 (eight tiles), and the
latency lists (one copy) beside them.   python profiles/factor_accuracy_outer_probe.py   (GPU box)"""
import sys
import numpy as np
from scipy.linalg import solve_triangular
sys.path.insert(0, '.')
from gpyrn_amd import _hip, covfunc   # noqa: E402
sys.path.insert(0, "profiles")
from factor_accuracy_probe import chol_ld, fsub_ld, padded   # noqa: E402
LD = np.longdouble
ctx = _hip.Context(0)
rng = np.random.RandomState(3)
N = 1000
t = np.sort(rng.uniform(0, 800, N))
r = t[:, None] - t[None, :]
for name, kern in (('Periodic', covfunc.Periodic(1.34, 22.7, 0.82)), ('Matern52', covfunc.Matern52(1.33, 32.0))):
    K = kern(r) + 1e-6 * np.eye(N)
    Lx = chol_ld(K)
    m = rng.standard_normal(N) * 10
    a = fsub_ld(Lx, m); ex = float(a @ a)
    la = solve_triangular(np.linalg.cholesky(K), m, lower=True)
    print('%-8s cond %.1e  m^T K^-1 m = %.6e  LAPACK %.1e' % (name, np.linalg.cond(K), ex, abs(float(la @ la) - ex) / ex), flush=True)
    for batch in (1, 5, 9):
        A = np.array([padded(K)] * batch)
        for acc in (0, 1):
            ctx.option('accurate_factor', acc)
            L, X, info = ctx.test_factor_invert(A)
            errs = []
            for b in range(batch):
                Xb = np.tril(X[b])[:N, :N]
                a = Xb.astype(LD) @ m.astype(LD)
                errs.append(abs(float(a @ a) - ex) / ex)
            a64 = np.tril(X[0])[:N, :N] @ m
            print('   batch %d (%s lists) %-12s: X m in long double %s | in float64 %.1e' % (
                batch, 'latency' if batch * 8 <= 32 else 'throughput', 'substitution' if acc else 'products',
                ' '.join('%.1e' % e for e in errs[:3]), abs(float(a64 @ a64) - ex) / ex), flush=True)
ctx.option('accurate_factor', -2)
