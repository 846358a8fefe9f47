"""Round 6: accuracy of the blocked factorisation on ill-conditioned PRIOR matrices, panel steps as products with explicit
inverses (option accurate_factor = 0: every round before this one) against panel steps by substitution (= 1; the default
for every factorisation of a prior matrix since round 6).  The library's own factor_invert through gprn_test_factor_invert;
"exact" is a long-double Cholesky + substitution of the same matrix, LAPACK is float64 cholesky + solve_triangular.

    python profiles/factor_accuracy_probe.py            (on the GPU box)
"""
import sys
import time

import numpy as np
from scipy.linalg import solve_triangular

sys.path.insert(0, '.')
from gpyrn_amd import _hip, covfunc   # noqa: E402

LD = np.longdouble


def chol_ld(K):
    n = K.shape[0]
    L = np.zeros((n, n), dtype=LD)
    A = K.astype(LD)
    for j in range(n):
        L[j, j] = np.sqrt(A[j, j] - np.dot(L[j, :j], L[j, :j]))
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def fsub_ld(L, b):
    n = L.shape[0]
    x = np.zeros(n, dtype=LD)
    for i in range(n):
        x[i] = (LD(b[i]) - np.dot(L[i, :i], x[:i])) / L[i, i]
    return x


def padded(K):
    n = K.shape[0]
    ld = (n + 127) // 128 * 128
    A = np.eye(ld)
    A[:n, :n] = K
    return A


def main():
    ctx = _hip.Context(0)
    rng = np.random.RandomState(3)
    kernels = (('Periodic', covfunc.Periodic(1.0, 17.0, 0.9)), ('SE', covfunc.SquaredExponential(1.0, 20.0)),
               ('QP', covfunc.QuasiPeriodic(1.0, 25.0, 17.0, 0.9)), ('Matern52', covfunc.Matern52(1.0, 30.0)),
               ('RQ', covfunc.RationalQuadratic(1.0, 0.7, 25.0)))
    for N in (100, 300, 1000):
        t = np.sort(rng.uniform(0, 80, N))
        r = t[:, None] - t[None, :]
        for name, kern in kernels:
            K = kern(r) + 1e-6 * np.eye(N)
            cond = np.linalg.cond(K)
            Lx = chol_ld(K)
            ld_exact = float(2 * np.sum(np.log(np.diag(Lx))))
            Xx = np.linalg.inv(Lx.astype(float))
            tr_exact = float(np.sum(Xx.astype(LD) ** 2))
            m_out = rng.standard_normal(N)
            m_in = K @ rng.standard_normal(N)
            ex = {}
            for mname, m in (('out', m_out), ('in', m_in)):
                a = fsub_ld(Lx, m)
                ex[mname] = float(a @ a)
            Ll = np.linalg.cholesky(K)
            row = ['%-8s N %4d cond %.1e' % (name, N, cond)]
            la = solve_triangular(Ll, m_out, lower=True)
            lap = (abs(float(la @ la) - ex['out']) / ex['out'],
                   abs(2 * np.sum(np.log(np.diag(Ll))) - ld_exact) / abs(ld_exact),
                   abs(float(np.sum(solve_triangular(Ll, np.eye(N), lower=True) ** 2)) - tr_exact) / tr_exact)
            row.append('LAPACK mKm %.1e logdet %.1e trKinv %.1e' % lap)
            for acc in (0, 1):
                ctx.option('accurate_factor', acc)
                L, X, info = ctx.test_factor_invert(padded(K))
                assert info == 0
                X = np.tril(X[0])[:N, :N]
                L = np.tril(L[0])[:N, :N]
                a = X.astype(LD) @ m_out.astype(LD)
                e_out = abs(float(a @ a) - ex['out']) / ex['out']
                a = X.astype(LD) @ m_in.astype(LD)
                e_in = abs(float(a @ a) - ex['in']) / ex['in']
                e_ld = abs(2 * np.sum(np.log(np.diag(L))) - ld_exact) / abs(ld_exact)
                e_tr = abs(float(np.sum(X.astype(LD) ** 2)) - tr_exact) / tr_exact
                row.append('| %s mKm out %.1e in %.1e logdet %.1e trKinv %.1e' % ('subst ' if acc else 'product', e_out, e_in, e_ld, e_tr))
            print(' '.join(row), flush=True)
    ctx.option('accurate_factor', -2)
    # what it costs: 8 matrices of 4096 (config 3's set-up) and 1 of 2048 (config 2's)
    for n, batch in ((4096, 8), (2048, 1), (512, 32)):
        t = np.sort(rng.uniform(0, 0.4 * n, n))
        A = np.array([np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 30.0 ** 2) + (1.0 + b) * np.eye(n) for b in range(batch)])
        for acc in (0, 1, 0, 1):
            ctx.option('accurate_factor', acc)
            ctx.test_factor_invert(A)
            ctx.profile_enable(('diag', 'panel', 'update', 'update_ahead'))
            t0 = time.perf_counter()
            ctx.test_factor_invert(A)
            wall = time.perf_counter() - t0
            pr = ctx.profile_read()
            ctx.profile_enable(())
            print('n %d batch %d accurate_factor %d: call %.1f ms (with copies); kernel families ms %s' % (
                n, batch, acc, wall * 1e3, {k: round(v[0], 3) for k, v in pr.items() if v[1]}), flush=True)
    ctx.option('accurate_factor', -2)
    ctx.close()


if __name__ == '__main__':
    main()
