import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests.test_parity_gpu import _random_problem
from oracle import cpu_ref
from scipy.linalg import solve_triangular
seed = 7
t, ys, es, nodes, weights, means, jit, p, q = _random_problem(seed)
y = np.array(ys)
Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, nodes, weights, means, jit, y)
mu0, var0 = cpu_ref.init_mu_var(y, [k.pars[0] for k in nodes], [k.pars[0] for k in weights], jit)
E, mu, var, pr = cpu_ref.sweep_B(Kf, Kw, Lf, Lw, yres, y, np.array(es)**2, j2, mu0, var0)
N = t.size
rows = mu.reshape(-1, N)
LD = np.longdouble
def chol_ld(K):
    L = np.zeros((N, N), dtype=LD); A = K.astype(LD)
    for j in range(N):
        L[j, j] = np.sqrt(A[j, j] - np.dot(L[j, :j], L[j, :j]))
        for i in range(j + 1, N):
            L[i, j] = (A[i, j] - np.dot(L[i, :j], L[j, :j])) / L[j, j]
    return L
def blocked(K, nb=16):
    """right-looking blocked Cholesky with explicit inverses of the diagonal blocks; X = L^-1 built alongside (LX = I)"""
    n = ((N + nb - 1) // nb) * nb
    A = np.eye(n); A[:N, :N] = K
    L = np.zeros((n, n)); X = np.zeros((n, n))
    R = np.eye(n)       # running right-hand side
    for k in range(0, n, nb):
        s = slice(k, k + nb)
        Lkk = np.linalg.cholesky(A[s, s]); Xkk = solve_triangular(Lkk, np.eye(nb), lower=True)
        L[s, s] = Lkk
        L[k + nb:, s] = A[k + nb:, s] @ Xkk.T
        X[s, :k + nb] = Xkk @ R[s, :k + nb]
        A[k + nb:, k + nb:] -= L[k + nb:, s] @ L[k + nb:, s].T
        R[k + nb:, :k + nb] -= L[k + nb:, s] @ X[s, :k + nb]
    return L[:N, :N], X[:N, :N]
kinds = [type(k).__name__ for k in nodes] + [type(k).__name__ for k in weights]
Ks = list(Kf) + list(Kw)
for gp, K in enumerate(Ks):
    m = rows[gp % rows.shape[0]]
    Lx = chol_ld(K)
    ax = solve_triangular(Lx.astype(float), m, lower=True)
    # exact in long double
    a = np.zeros(N, dtype=LD)
    for i in range(N):
        a[i] = (LD(m[i]) - np.dot(Lx[i, :i], a[:i])) / Lx[i, i]
    exact = float(a @ a)
    Ll = np.linalg.cholesky(K); al = solve_triangular(Ll, m, lower=True); lap = float(al @ al)
    Lb, Xb = blocked(K)
    ab = Xb @ m; bx = float(ab @ ab)
    abl = solve_triangular(Lb, m, lower=True); bl = float(abl @ abl)
    Xl = solve_triangular(Ll, np.eye(N), lower=True); axl = Xl @ m; xl = float(axl @ axl)
    print('%-18s exact %.4e |m|^2 %.2e  LAPACK solve %.1e | LAPACK explicit inverse x m %.1e | blocked-16: X m %.1e, L solve %.1e' % (
        kinds[gp], exact, m @ m, abs(lap - exact) / exact, abs(xl - exact) / exact, abs(bx - exact) / exact, abs(bl - exact) / exact))
