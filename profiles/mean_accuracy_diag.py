"""Round 6: where the posterior MEAN of an ill-conditioned latent GP loses its digits.  The illc_N1000_p2q3 fixture: LogP off by
1.3e-8 although m^T K^-1 m is formed accurately from the device's state -- the state's own error (3.7e-11 norm-wise on the pure
Periodic weight's mean) is amplified by K^-1.  For that latent GP after the first sweep: its d, pred, B = I + D^1/2 K D^1/2 on
the host from the device's state; the mean from (a) LAPACK's factor of B and a substitution-built X, (b) the device's own X
(read back) in a long-double product, (c) the device's mean; all against a long-double evaluation.
    python profiles/mean_accuracy_diag.py [tag] [gp]"""
import sys
import numpy as np
from scipy.linalg import solve_triangular
sys.path.insert(0, '.')
sys.path.insert(0, 'profiles')
import gpyrn_amd as gpyrn   # noqa: E402
from gpyrn_amd import _hip, covfunc, meanfunc   # noqa: E402
from oracle import cpu_ref   # noqa: E402
from tests import _cases   # noqa: E402
from factor_accuracy_probe import chol_ld   # noqa: E402
LD = np.longdouble
tag = sys.argv[1] if len(sys.argv) > 1 else 'illc_N1000_p2q3'
gp = int(sys.argv[2]) if len(sys.argv) > 2 else 6
meta, d_ = _cases.load(tag)
p, q, N = meta['p'], meta['q'], meta['N']
nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
g = gpyrn.inference(q, np.array(d_['time']), *_cases.data_args(d_))
g.set_components(nodes, weights, means, jit)
for acc in (-2, 1):
    g._backend().option('accurate_factor', acc)
    g._prior_key = None
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    ctx.set_muvar(d_['mu_init'], d_['var_init'])
    ctx.sweep(1, commit=True)
    mu, var = ctx.get_muvar()
    j, i = divmod(gp - q, p)
    row = (1 + i) * q + j
    Kf, Kw, Lf, Lw, y, j2 = cpu_ref.setup(d_['time'], nodes, weights, means, jit, d_['y'])
    variance = j2[:, None] + d_['yerr'] ** 2
    mu3, var3 = mu.reshape(p + 1, q, N), var.reshape(p + 1, q, N)
    muW_old = np.asarray(d_['mu_init']).reshape(p + 1, q, N)[1:]
    dd, pred = cpu_ref._weight_d_and_pred(y, variance, mu3[0], var3[0], muW_old, j, i)
    K = ctx.get_matrix(_hip.M_K, gp)
    s = np.sqrt(dd)
    B = K * s[:, None] * s[None, :] + np.eye(N)
    z = pred / s
    # exact (long double)
    Lx = chol_ld(B)
    Xx = np.zeros((N, N), dtype=LD)
    for c in range(N):
        e = np.zeros(N, dtype=LD); e[c] = 1
        for r in range(c, N):
            e[r] = (e[r] - np.dot(Lx[r, c:r], e[c:r])) / Lx[r, r]
        Xx[:, c] = e
    zz = z.astype(LD)
    m_exact = ((zz - Xx.T @ (Xx @ zz)) / s.astype(LD)).astype(float)
    Ll = np.linalg.cholesky(B)
    Xl = solve_triangular(Ll, np.eye(N), lower=True)
    m_lap = (z - Xl.T @ (Xl @ z)) / s
    Xd = np.tril(ctx.get_matrix(_hip.M_BX, gp))
    m_devX_ld = (((zz - Xd.astype(LD).T @ (Xd.astype(LD) @ zz)) / s.astype(LD))).astype(float)
    m_devX_64 = (z - Xd.T @ (Xd @ z)) / s
    m_dev = mu3[1 + i, j]
    sc = np.abs(m_exact).max()
    w = np.linalg.eigvalsh(B)
    Kl = np.linalg.cholesky(K)
    def mkm(m):
        a = solve_triangular(Kl, m, lower=True); return float(a @ a)
    print('accurate_factor %d, latent GP %d (%s): min d %.2e max d %.2e cond(B) %.1e' % (acc, gp, type((nodes + weights)[gp]).__name__, dd.min(), dd.max(), w[-1] / w[0]))
    for name, m in (('LAPACK factor, X by substitution', m_lap), ("device's X, long-double products", m_devX_ld), ("device's X, float64 products", m_devX_64), ("device's mean", m_dev)):
        print('   %-34s mean off by %.1e (norm-wise)   m^T K^-1 m off by %.1e' % (name, np.abs(m - m_exact).max() / sc, abs(mkm(m) - mkm(m_exact)) / mkm(m_exact)))
    print('   X itself: max |X_dev - X_exact| / max |X_exact| = %.1e ; LAPACK substitution %.1e' % (
        np.abs(Xd - Xx.astype(float)).max() / np.abs(Xx).max(), np.abs(Xl - Xx.astype(float)).max() / np.abs(Xx).max()), flush=True)
