import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gpyrn_amd import covfunc
from scipy.linalg import solve_triangular
LD = np.longdouble
rng = np.random.RandomState(3)
N = 300
t = np.sort(rng.uniform(0, 80, N))
def chol_ld(K):
    n = K.shape[0]; L = np.zeros((n, n), dtype=LD); A = K.astype(LD)
    for j in range(n):
        L[j, j] = np.sqrt(A[j, j] - np.dot(L[j, :j], L[j, :j]))
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - np.dot(L[i, :j], L[j, :j])) / L[j, j]
    return L
def inner(A, nb=16, refine=0):
    """factor + inverse of one tile by 16-blocks with explicit inverses"""
    n = A.shape[0]; A = A.copy(); L = np.zeros((n, n)); X = np.zeros((n, n)); R = np.eye(n)
    for k in range(0, n, nb):
        s = slice(k, k + nb)
        Lkk = np.linalg.cholesky(A[s, s]); Xkk = solve_triangular(Lkk, np.eye(nb), lower=True)
        L[s, s] = Lkk
        P = A[k + nb:, s] @ Xkk.T
        for _ in range(refine): P = P + (A[k + nb:, s] - P @ Lkk.T) @ Xkk.T
        L[k + nb:, s] = P
        X[s, :k + nb] = Xkk @ R[s, :k + nb]
        A[k + nb:, k + nb:] -= P @ P.T
        R[k + nb:, :k + nb] -= P @ X[s, :k + nb]
    return L, X
def outer(K, refine_in=0, refine_out=0, nb=128):
    n = ((K.shape[0] + nb - 1) // nb) * nb
    A = np.eye(n); A[:K.shape[0], :K.shape[0]] = K
    L = np.zeros((n, n)); X = np.zeros((n, n)); R = np.eye(n)
    for k in range(0, n, nb):
        s = slice(k, k + nb)
        Lkk, Xkk = inner(A[s, s], 16, refine_in)
        L[s, s] = Lkk
        P = A[k + nb:, s] @ Xkk.T
        for _ in range(refine_out): P = P + (A[k + nb:, s] - P @ Lkk.T) @ Xkk.T
        L[k + nb:, s] = P
        X[s, :k + nb] = Xkk @ R[s, :k + nb]
        A[k + nb:, k + nb:] -= P @ P.T
        R[k + nb:, :k + nb] -= P @ X[s, :k + nb]
    m = K.shape[0]
    return L[:m, :m], X[:m, :m]
for name, kern in (('Periodic', covfunc.Periodic(1.0, 17.0, 0.9)), ('SE', covfunc.SquaredExponential(1.0, 20.0)), ('QP', covfunc.QuasiPeriodic(1.0, 25.0, 17.0, 0.9))):
    r = t[:, None] - t[None, :]
    K = kern(r) + 1e-6 * np.eye(N)
    Lx = chol_ld(K)
    for mname, m in (('random m', rng.standard_normal(N)), ('m = K z', K @ rng.standard_normal(N))):
        a = np.zeros(N, dtype=LD)
        for i in range(N): a[i] = (LD(m[i]) - np.dot(Lx[i, :i], a[:i])) / Lx[i, i]
        exact = float(a @ a)
        Ll = np.linalg.cholesky(K); al = solve_triangular(Ll, m, lower=True)
        res = []
        for ri, ro in ((0, 0), (1, 0), (0, 1), (1, 1)):
            Lb, Xb = outer(K, ri, ro)
            ab = Xb @ m
            res.append(abs(float(ab @ ab) - exact) / exact)
        print('%-9s cond %.1e %-9s exact %.3e | LAPACK %.1e | two-level plain %.1e | 16-level refined %.1e | 128-level refined %.1e | both %.1e' % (
            name, np.linalg.cond(K), mname, exact, abs(float(al @ al) - exact) / exact, *res))
