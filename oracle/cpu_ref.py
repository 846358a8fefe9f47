"""CPU oracle for the gpyrn mean-field ELBO hot path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module; gpyrn_amd never does (the product path has no CPU
fallback).  Everything is fp64 NumPy/SciPy (LAPACK), like the reference.

Two restatements of one sweep (= one `ELBOaux` call, meanfield.py:651-710):

* form "ref"  -- the reference's own formulation, quirk for quirk: Woodbury with
  a general LU solve (meanfield.py:771,850), Cholesky of the explicit Sigma
  for the entropy (:1087-1090), cho_solve traces for the prior (:1041,1051).
  7 N^3 flop per latent GP.  This is what bench.py times as `cpu_baseline`.
* form "B"    -- the algebra the HIP path executes: B = I + D^1/2 K D^1/2 =
  L L^T, X = L^-1; diag Sigma = (1 - colnorm2(X)) / d, log det Sigma =
  log det K - log det B, tr(K^-1 Sigma) = tr(B^-1), Sigma v = D^-1/2 (q - X^T X q), q = D^-1/2 v.
  2/3 N^3 per GP (+1/3 N^3 for the cumulative-trace quirk Q1 when q >= 2).

Pinned against tests/golden/*.npz, which oracle/gen_golden.py produced by
running the reference itself (tests/test_oracle.py).  Quirks kept (SURVEY.md
§8a): Q1 cumulative sumSigmaF, Q2 mu_w raw reshape, Q3 raw y in the likelihood,
Q4 _initMuVar layout, Q5 ELBO/q, Q6 Jacobi ordering, Q7 discarded first sweep.
"""
import numpy as np
from scipy.linalg import cho_solve, lu_factor, lu_solve, solve_triangular

NUGGET = 1e-6          # meanfield.py:433
LOG2PI = np.log(2 * np.pi)


# ------------------------------------------------------------------ layout
def split_u(u, p, q, N):
    """meanfield.py:473-489: first q*N entries are the nodes, the rest is
    *reshaped* (not transposed) to (p, q, N)."""
    u = np.asarray(u, dtype=float).ravel()
    return u[:q * N].reshape(q, N), u[q * N:].reshape(p, q, N)


def init_mu_var(y, node_amp, weight_amp, jitters):
    """meanfield.py:491-510.  `y` is the raw (p, N) data.  Only the first p
    weight amplitudes are ever used (zip truncation), and the weight block is
    emitted node-major although split_u reads it output-major (Q4)."""
    y = np.asarray(y, dtype=float)
    p, N = y.shape
    jit = np.asarray(jitters, dtype=float)
    wamp = np.asarray(weight_amp, dtype=float)[:p]
    mu_f, mu_w, var_f, var_w = [], [], [], []
    for a in node_amp:
        m = np.sqrt(np.abs(y) * a / wamp[:, None]) * np.sign(y)
        mu_f.append(m.mean(axis=0))
        mu_w.append(np.sqrt(np.abs(y) * wamp[:, None] / a))
        var_f.append(np.full(N, jit.mean()))
        var_w.append(np.repeat(jit[:, None], N, axis=1))
    mu = np.concatenate([np.ravel(mu_f), np.ravel(mu_w)])
    var = np.concatenate([np.ravel(var_f), np.ravel(var_w)])
    return mu, var


def kmatrix(kernel, time):
    """meanfield.py:413-434 for one-argument kernels."""
    r = time[:, None] - time[None, :]
    return kernel(r) + NUGGET * np.eye(time.size)


# ------------------------------------------------- shared O(pqN) pieces
def _node_d_and_pred(y, variance, muF, muW, varW, j):
    """d_j and the right-hand side of the node-mean update, meanfield.py:765-791."""
    d = np.sum((muW[:, j] ** 2 + varW[:, j]) / variance, axis=0)
    others = np.delete(np.arange(muF.shape[0]), j)
    resid = y - np.sum(muW[:, others] * muF[others][None], axis=1)
    pred = np.sum(resid * muW[:, j] / variance, axis=0)
    return d, pred


def _weight_d_and_pred(y, variance, mu_f_new, diag_sf, muW_old, j, i):
    """meanfield.py:838-864 (new mu_f, old mu_w)."""
    d = (mu_f_new[j] ** 2 + diag_sf[j]) / variance[i]
    others = np.delete(np.arange(mu_f_new.shape[0]), j)
    resid = y[i] - np.sum(mu_f_new[others] * muW_old[i, others], axis=0)
    pred = resid * mu_f_new[j] / variance[i]
    return d, pred


def expected_loglike(y_raw, variance, mu_f, mu_w, dsf, dsw):
    """meanfield.py:895-990.  y_raw is the *raw* data (Q3); dsw is (q, p, N)."""
    p, q, N = mu_w.shape
    out = -0.5 * np.sum(np.log(2 * np.pi * variance))
    fit = np.einsum('iqn,qn->in', mu_w, mu_f)
    out += -0.5 * np.sum((y_raw - fit) ** 2 / variance)
    acc = 0.0
    for i in range(p):
        for j in range(q):
            acc += np.dot(dsf[j], mu_w[i, j] ** 2 / variance[i])
            acc += np.dot(dsw[j, i], mu_f[j] ** 2 / variance[i])
            acc += np.dot(dsf[j], dsw[j, i] / variance[i])
    return out - 0.5 * acc


# --------------------------------------------------------- form "ref"
def sweep_ref(Kf, Kw, Lf, Lw, y, y_raw, yerr2, jitt2, mu, var, return_sigma=False):
    """One ELBOaux in the reference's formulation.

    Kf (q,N,N), Kw (q*p,N,N) flat index j*p+i, Lf/Lw their lower Cholesky
    factors, y (p,N) mean-subtracted, y_raw (p,N).  Returns
    (ELBO, new_mu (p+1,q,N), new_var (p+1,q,N), parts (LogL, LogP, Ent)).
    """
    q, N = Kf.shape[0], Kf.shape[-1]
    p = Kw.shape[0] // q
    Kw4 = Kw.reshape(q, p, N, N)
    Lw4 = Lw.reshape(q, p, N, N)
    muF, muW = split_u(mu, p, q, N)
    varF, varW = split_u(var, p, q, N)
    variance = jitt2[:, None] + yerr2

    sig_f = np.empty((q, N, N))
    mu_f = np.empty((q, N))
    for j in range(q):
        d, pred = _node_d_and_pred(y, variance, muF, muW, varW, j)
        lu = lu_factor(np.diag(1.0 / d) + Kf[j])
        sig_f[j] = Kf[j] - Kf[j] @ lu_solve(lu, Kf[j])
        mu_f[j] = sig_f[j] @ pred
    dsf = np.einsum('jnn->jn', sig_f).copy()

    sig_w = np.empty((q, p, N, N))
    mu_w = np.empty((p, q, N))
    for j in range(q):
        for i in range(p):
            d, pred = _weight_d_and_pred(y, variance, mu_f, dsf, muW, j, i)
            lu = lu_factor(np.diag(1.0 / d) + Kw4[j, i])
            sig_w[j, i] = Kw4[j, i] - Kw4[j, i] @ lu_solve(lu, Kw4[j, i])
            mu_w[i, j] = sig_w[j, i] @ pred
    dsw = np.einsum('jinn->jin', sig_w).copy()

    # entropy, meanfield.py:1069-1093
    ent = 0.0
    for j in range(q):
        ent += np.sum(np.log(np.diag(np.linalg.cholesky(sig_f[j]))))
        for i in range(p):
            ent += np.sum(np.log(np.diag(np.linalg.cholesky(sig_w[j, i]))))
    ent += 0.5 * q * (p + 1) * N * (1 + LOG2PI)

    # expected log prior, meanfield.py:992-1067 (Q1, Q2)
    m_scr = mu_w.reshape(q, p, N)
    logp = 0.0
    cum = np.zeros((N, N))
    for j in range(q):
        cum = cum + sig_f[j]
        a = cho_solve((Lf[j], True), mu_f[j])
        tr = np.trace(cho_solve((Lf[j], True), cum))
        logp += -np.sum(np.log(np.diag(Lf[j]))) - 0.5 * (mu_f[j] @ a + tr)
        for i in range(p):
            a = cho_solve((Lw4[j, i], True), m_scr[j, i])
            tr = np.trace(cho_solve((Lw4[j, i], True), sig_w[j, i]))
            logp += -np.sum(np.log(np.diag(Lw4[j, i]))) - 0.5 * (m_scr[j, i] @ a + tr)
    logp += -0.5 * N * q * (p + 1) * LOG2PI

    logl = expected_loglike(y_raw, variance, mu_f, mu_w, dsf, dsw)
    new_mu = np.concatenate((mu_f[None], mu_w))
    new_var = np.concatenate((dsf[None], np.transpose(dsw, (1, 0, 2))))
    if return_sigma:
        return (logl + logp + ent) / q, new_mu, new_var, (logl, logp, ent), sig_f, sig_w
    return (logl + logp + ent) / q, new_mu, new_var, (logl, logp, ent)


def fixed_state_elbo(Kf, Kw, y_raw, yerr2, jitt2, mu_f, mu_w, sig_f, sig_w):
    """(LogL + LogP)/q of meanfield.py:895-1067 as a function of the prior matrices and the jitters, with the
    variational means (mu_f (q,N), mu_w (p,q,N)) and covariances (sig_f (q,N,N), sig_w (q,p,N,N)) held fixed --
    the entropy does not depend on either.  Finite differences of this are what the analytic gradient of
    gpyrn_amd.inference.grad_ELBO is checked against."""
    q, N = Kf.shape[0], Kf.shape[-1]
    p = Kw.shape[0] // q
    Kw4 = Kw.reshape(q, p, N, N)
    variance = jitt2[:, None] + yerr2
    m_scr = mu_w.reshape(q, p, N)
    logp = 0.0
    cum = np.zeros((N, N))
    for j in range(q):
        cum = cum + sig_f[j]
        L = np.linalg.cholesky(Kf[j])
        logp += -np.sum(np.log(np.diag(L))) - 0.5 * (mu_f[j] @ cho_solve((L, True), mu_f[j])
                                                     + np.trace(cho_solve((L, True), cum)))
        for i in range(p):
            L = np.linalg.cholesky(Kw4[j, i])
            logp += -np.sum(np.log(np.diag(L))) - 0.5 * (m_scr[j, i] @ cho_solve((L, True), m_scr[j, i])
                                                         + np.trace(cho_solve((L, True), sig_w[j, i])))
    dsf = np.einsum('jnn->jn', sig_f)
    dsw = np.einsum('jinn->jin', sig_w)
    logl = expected_loglike(y_raw, variance, mu_f, mu_w, dsf, dsw)
    return (logl + logp) / q


# ----------------------------------------------------------- form "B"
def _gp_update_B(K, d, pred, need_inverse=False):
    """One latent GP in B-form.  Returns diag Sigma, Sigma@pred, log det B,
    tr(B^-1), and optionally the explicit B^-1 and s = sqrt(d)."""
    s = np.sqrt(d)
    B = K * s[:, None] * s[None, :]
    B[np.diag_indices_from(B)] += 1.0
    L = np.linalg.cholesky(B)
    X = solve_triangular(L, np.eye(L.shape[0]), lower=True)
    binv_diag = np.sum(X * X, axis=0)
    q = pred / s                          # Sigma pred = D^-1/2 (I - B^-1) D^-1/2 pred
    sig_pred = (q - X.T @ (X @ q)) / s
    logdetB = 2.0 * np.sum(np.log(np.diag(L)))
    Binv = X.T @ X if need_inverse else None
    return (1.0 - binv_diag) / d, sig_pred, logdetB, np.sum(binv_diag), Binv, s


def sweep_B(Kf, Kw, Lf, Lw, y, y_raw, yerr2, jitt2, mu, var, Kf_inv=None):
    """Same contract as sweep_ref, in the algebra of the HIP path."""
    q, N = Kf.shape[0], Kf.shape[-1]
    p = Kw.shape[0] // q
    Kw4 = Kw.reshape(q, p, N, N)
    Lw4 = Lw.reshape(q, p, N, N)
    muF, muW = split_u(mu, p, q, N)
    varF, varW = split_u(var, p, q, N)
    variance = jitt2[:, None] + yerr2
    if Kf_inv is None and q > 1:
        Kf_inv = [None] + [cho_solve((Lf[j], True), np.eye(N)) for j in range(1, q)]

    ent = 0.5 * q * (p + 1) * N * (1 + LOG2PI)
    logp = -0.5 * N * q * (p + 1) * LOG2PI
    mu_f = np.empty((q, N))
    dsf = np.empty((q, N))
    sig_parts = []                       # (Binv_k, s_k) of earlier nodes, for Q1
    for j in range(q):
        d, pred = _node_d_and_pred(y, variance, muF, muW, varW, j)
        dsf[j], mu_f[j], ldB, trBinv, Binv, s = _gp_update_B(
            Kf[j], d, pred, need_inverse=(j < q - 1))
        ldK = 2.0 * np.sum(np.log(np.diag(Lf[j])))
        ent += 0.5 * (ldK - ldB)
        tr = trBinv
        for (Bk, sk) in sig_parts:       # Q1: + tr(K_j^-1 Sigma_k), k < j
            Sk = (np.eye(N) - Bk) / (sk[:, None] * sk[None, :])
            tr += np.sum(Kf_inv[j] * Sk)
        if Binv is not None:
            sig_parts.append((Binv, s))
        a = solve_triangular(Lf[j], mu_f[j], lower=True)
        logp += -0.5 * ldK - 0.5 * (a @ a + tr)

    mu_w = np.empty((p, q, N))
    dsw = np.empty((q, p, N))
    ldKw = np.empty((q, p))
    trw = np.empty((q, p))
    for j in range(q):
        for i in range(p):
            d, pred = _weight_d_and_pred(y, variance, mu_f, dsf, muW, j, i)
            dsw[j, i], mu_w[i, j], ldB, trw[j, i], _, _ = _gp_update_B(Kw4[j, i], d, pred)
            ldKw[j, i] = 2.0 * np.sum(np.log(np.diag(Lw4[j, i])))
            ent += 0.5 * (ldKw[j, i] - ldB)
    m_scr = mu_w.reshape(q, p, N)        # Q2
    for j in range(q):
        for i in range(p):
            a = solve_triangular(Lw4[j, i], m_scr[j, i], lower=True)
            logp += -0.5 * ldKw[j, i] - 0.5 * (a @ a + trw[j, i])

    logl = expected_loglike(y_raw, variance, mu_f, mu_w, dsf, dsw)
    new_mu = np.concatenate((mu_f[None], mu_w))
    new_var = np.concatenate((dsf[None], np.transpose(dsw, (1, 0, 2))))
    return (logl + logp + ent) / q, new_mu, new_var, (logl, logp, ent)


# ------------------------------------------------------------- driver
def setup(time, nodes, weights, means, jitters, y_raw):
    """The setup block of ELBOcalc, meanfield.py:618-624.  nodes/weights are
    callables on an (N,N) difference matrix; means are callables or None."""
    Kf = np.array([kmatrix(k, time) for k in nodes])
    Kw = np.array([kmatrix(k, time) for k in weights])
    Lf = np.array([np.linalg.cholesky(K) for K in Kf])
    Lw = np.array([np.linalg.cholesky(K) for K in Kw])
    m = np.array([np.zeros_like(time) if f is None else f(time) for f in means])
    jitt2 = np.asarray(jitters, dtype=float) ** 2
    return Kf, Kw, Lf, Lw, np.asarray(y_raw) - m, jitt2


def elbo_calc(Kf, Kw, Lf, Lw, y, y_raw, yerr2, jitt2, mu, var,
              max_iter=10000, form='ref'):
    """ELBOcalc's loop and stop rule, meanfield.py:626-649.
    Returns (ELBO, mu, var, iterNumber, elboArray)."""
    sweep = sweep_ref if form == 'ref' else sweep_B
    E, *_ = sweep(Kf, Kw, Lf, Lw, y, y_raw, yerr2, jitt2, mu, var)   # Q7
    hist = [E]
    it = 0
    while it < max_iter:
        E, mu, var, _ = sweep(Kf, Kw, Lf, Lw, y, y_raw, yerr2, jitt2, mu, var)
        hist.append(E)
        it += 1
        if it > 3:
            last = np.array(hist[-3:])
            crit = np.abs(np.std(last) / np.mean(last))
            if crit < 1e-3 and crit != 0:
                break
    return E, mu, var, it, np.array(hist)


# --------------------------------------------------------- prediction
def gp_prediction(kernel, time, tstar, m, v):
    """_gp.GP.prediction (_gp.py:107-138): conditional mean/variance of one latent GP at tstar,
    with the 1.25e-12 nugget of _gp.py:47 (one-argument kernels)."""
    r = time[:, None] - time[None, :]
    cov = kernel(r) + 1.25e-12 * np.eye(time.size) + np.diag(v)
    L = np.linalg.cholesky(cov)
    Ks = kernel(tstar[:, None] - time[None, :])
    sol = cho_solve((L, True), m)
    W = solve_triangular(L, Ks.T, lower=True)
    kss = kernel(np.zeros(tstar.size)) + 1.25e-12
    return Ks @ sol, kss - np.sum(W * W, axis=0)


def prediction(time, tstar, nodes, weights, means, jitters, mu, var, p, q):
    """inference._Prediction (meanfield.py:1289-1381); returns (mean (N*,p), var (N*,p), node
    means (q,N*), weight means (q*p,N*))."""
    N = time.size
    muF, muW = split_u(mu, p, q, N)
    varF, varW = split_u(var, p, q, N)
    nP, nV, wP, wV = [], [], [], []
    for j in range(q):
        a, b = gp_prediction(nodes[j], time, tstar, muF[j], varF[j])
        nP.append(a); nV.append(b)
        for i in range(p):
            a, b = gp_prediction(weights[j * p + i], time, tstar, muW[i, j], varW[i, j])
            wP.append(a); wV.append(b)
    nP, nV, wP, wV = map(np.array, (nP, nV, wP, wV))
    wPq, wVq = wP.reshape(q, p, -1), wV.reshape(q, p, -1)
    jitt2 = np.asarray(jitters, dtype=float) ** 2
    mean = np.zeros((tstar.size, p))
    variance = np.zeros((tstar.size, p))
    for i in range(p):
        mean[:, i] += np.zeros(tstar.size) if means[i] is None else means[i](tstar)
        for j in range(q):
            mean[:, i] += nP[j] * wPq[j, i]
            variance[:, i] += wPq[j, i] ** 2 * nV[j] + wVq[j, i] * (nV[j] + nP[j] ** 2) + jitt2[i]
    return mean, variance, nP, wP
