import sys, numpy as np
sys.path.insert(0, '.')
import gpyrn_amd as gpyrn
from gpyrn_amd import _hip
from tests.test_parity_gpu import _random_problem
from scipy.linalg import solve_triangular
# argument: a seed of tests/test_parity_gpu.py::_random_problem (default 7), or the tag of a golden fixture (e.g. illc_N1000_p2q3)
arg = sys.argv[1] if len(sys.argv) > 1 else '7'
if arg.lstrip('-').isdigit():
    t, ys, es, nodes, weights, means, jit, p, q = _random_problem(int(arg))
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
else:
    from tests import _cases
    from gpyrn_amd import covfunc, meanfunc
    meta, d_ = _cases.load(arg)
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    p, q, t = meta['p'], meta['q'], np.array(d_['time'])
    g = gpyrn.inference(q, t, *_cases.data_args(d_))
    g.set_components(nodes, weights, means, jit)
ctx = g._setup_device(nodes, weights, means, jit)
mu0, var0 = g._initMuVar(nodes, weights, jit)
ctx.set_muvar(mu0, var0)
ctx.sweep(1, commit=True)
mu, var = ctx.get_muvar()
N = t.size
LD = np.longdouble


def chol_ld(K):
    L = np.zeros((N, N), dtype=LD)
    A = K.astype(LD)
    for j in range(N):
        s = A[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(s)
        if j + 1 < N:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def fsub_ld(L, b):
    x = np.zeros(N, dtype=LD)
    for i in range(N):
        x[i] = (b[i].astype(LD) if hasattr(b[i], 'astype') else LD(b[i])) - np.dot(L[i, :i], x[:i])
        x[i] /= L[i, i]
    return x


rows = mu.reshape(-1, N)
kinds = [type(k).__name__ for k in nodes] + [type(k).__name__ for k in weights]
for gp in range(q * (p + 1)):
    K = ctx.get_matrix(_hip.M_K, gp)
    Xd = ctx.get_matrix(_hip.M_KLINV, gp)
    m = rows[gp % rows.shape[0]]
    w = np.linalg.eigvalsh(K)
    Lx = chol_ld(K)
    exact = float(np.dot(fsub_ld(Lx, m), fsub_ld(Lx, m)))
    Ll = np.linalg.cholesky(K)
    a = solve_triangular(Ll, m, lower=True)
    lap = float(a @ a)
    ad = Xd.astype(LD) @ m.astype(LD)
    dev_x = float(ad @ ad)
    dev = ctx.prior_terms(gp, np.eye(N), m)
    ld_exact = float(2 * np.sum(np.log(np.diag(Lx))))
    print('%-18s cond %.1e | m^T K^-1 m exact %.6e | LAPACK %.1e | device X (long-double product) %.1e | device %.1e || logdet: LAPACK %.1e device %.1e | tr K^-1: LAPACK %.1e device %.1e' % (
        kinds[gp], w[-1] / w[0], exact, abs(lap - exact) / exact, abs(dev_x - exact) / exact, abs(dev[1] - exact) / exact,
        abs(2 * np.sum(np.log(np.diag(Ll))) - ld_exact) / abs(ld_exact), abs(dev[0] - ld_exact) / abs(ld_exact),
        abs(np.trace(np.linalg.inv(K)) - float(np.sum(np.linalg.inv(Lx.astype(float))**2))) / float(np.sum(np.linalg.inv(Lx.astype(float))**2)),
        abs(dev[2] - float(np.sum(np.linalg.inv(Lx.astype(float))**2))) / float(np.sum(np.linalg.inv(Lx.astype(float))**2))))
