"""Where do the device's posterior variances lose their digits?  (VERDICT r4 weak #1 / next #2.)

var = (1 - sum_r X_rc^2) / d with X = chol(B)^-1, B = I + D^1/2 K D^1/2.  For the fixture `tag` (default
cfg5shape_N1024: p = 4, q = 3, two forced sweeps) this runs the sweeps on the GPU, reads the device's X of every latent
GP back (gprn_get_matrix, GPRN_M_BX) and recomputes the variances on the host from
  (a) the device's X, column sums in long double       -> the device's summation is not in it
  (b) the device's X, column sums in fp64 (NumPy)
  (c) LAPACK's X for the same B (potrf + trtri), sums in long double   -> the device's X is not in it
each against the reference's own variances (tests/golden), norm-wise per latent GP as tests/_cases.assert_state does, next to
what the device itself returned.  `python profiles/var_accuracy_diag.py [tag]` on a GPU box."""
import sys

import numpy as np
from scipy.linalg import solve_triangular

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import gpyrn_amd as gpyrn                                    # noqa: E402
from gpyrn_amd import _hip, covfunc, meanfunc                # noqa: E402
from tests import _cases                                     # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'cfg5shape_N1024'
meta, d = _cases.load(tag)
nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
g = gpyrn.inference(meta['q'], np.array(d['time']), *_cases.data_args(d))
g.set_components(nodes, weights, means, jit)
q, p, N, n = meta['q'], meta['p'], meta['N'], int(meta['nsweeps'])
mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
ctx.set_muvar(mu0, var0)
if n > 1:
    ctx.sweep(n - 1, commit=True)
mu_p, var_p = ctx.get_muvar()                                  # the state the last sweep starts from
ctx.sweep(1, commit=True)
mu_f, var_f = ctx.get_muvar()
ref = np.asarray(d['var_final']).reshape(p + 1, q, N)
variance = np.asarray(jit, dtype=float)[:, None] ** 2 + g.yerr2


def rowerr(a, r):
    return float(np.abs(a - r).max() / np.abs(r).max())


print('%s: N = %d, p = %d, q = %d, %d forced sweeps; deviation of the variances from the reference, norm-wise per latent GP' % (tag, N, p, q, n))
print('%-10s %9s | %10s %10s %10s %10s | %9s %9s' % ('latent GP', 'min d', 'device', '(a) X ld', '(b) X f64', '(c) LAPACK', '|X-Xl|/|Xl|', 'cond B'))
worst = np.zeros(4)
for gp in range(q * (p + 1)):
    if gp < q:
        j = gp
        dd = sum((mu_p[1 + i, j] ** 2 + var_p[1 + i, j]) / variance[i] for i in range(p))
        row = (0, j)
        K = g._KMatrix(g.nodes[j])
    else:
        j, i = divmod(gp - q, p)
        dd = (mu_f[0, j] ** 2 + var_f[0, j]) / variance[i]
        row = (1 + i, j)
        K = g._KMatrix(g.weights[gp - q])
    X = ctx.get_matrix(_hip.M_BX, gp)
    s = np.sqrt(dd)
    B = K * s[:, None] * s[None, :] + np.eye(N)
    L = np.linalg.cholesky(B)
    Xl = solve_triangular(L, np.eye(N), lower=True)
    r = ref[row]
    dev = rowerr(var_f[row], r)
    a = rowerr(np.asarray((1.0 - (X.astype(np.longdouble) ** 2).sum(axis=0)) / dd, dtype=float), r)
    b = rowerr((1.0 - np.sum(X * X, axis=0)) / dd, r)
    c_ = rowerr(np.asarray((1.0 - (Xl.astype(np.longdouble) ** 2).sum(axis=0)) / dd, dtype=float), r)
    worst = np.maximum(worst, [dev, a, b, c_])
    print('%-10s %9.2e | %10.2e %10.2e %10.2e %10.2e | %9.2e %9.2e' % (
        ('node %d' % gp) if gp < q else 'weight %d,%d' % (j, i), dd.min(), dev, a, b, c_,
        np.abs(X - Xl).max() / np.abs(Xl).max(), np.linalg.cond(B)))
print('%-10s %9s | %10.2e %10.2e %10.2e %10.2e' % ('worst', '', *worst))
