#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

TEST INFRASTRUCTURE -- runs only in the build container, where
/root/reference exists.  It imports the reference package read-only, with the
NumPy/SciPy stand-ins of oracle/standins/ for the two absent third-party
modules (jax, emcee), feeds it seeded inputs and stores
inputs + outputs as small .npz/.json fixtures.  Nothing here is imported by
gpyrn_amd, and the GPU box never sees the reference in any form.

    python oracle/gen_golden.py [--big]     # --big adds N=2048 and N=4096 cases

Reference entry points exercised (all in /root/reference/gpyrn):
  covfunc.*.__call__ via inference._KMatrix          meanfield.py:413-434
  inference._initMuVar / _u_to_fhatW                 meanfield.py:473-510
  inference._updateSigMu/_entropy/_expectedLogPrior/_expectedLogLike
                                                     meanfield.py:713-1093
  inference.ELBOaux / ELBOcalc                       meanfield.py:561-710
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get('GPYRN_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, 'standins'))
os.environ.setdefault('MPLBACKEND', 'Agg')

import numpy as np  # noqa: E402

if not hasattr(np, 'float'):
    np.float = float  # meanfield.py:177 uses the removed alias

import gpyrn  # noqa: E402  (the reference)
from gpyrn import covfunc as rcov, meanfunc as rmean  # noqa: E402
from gpyrn.meanfield import inference as rinference  # noqa: E402

sys.path.insert(0, REPO)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    'synth', os.path.join(REPO, 'gpyrn_amd', 'synth.py'))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


# ----------------------------------------------------------------- kernels
KERNEL_CASES = [
    ('Constant', [1.3]),
    ('WhiteNoise', [0.7]),
    ('SquaredExponential', [1.2, 7.5]),
    ('Periodic', [0.9, 11.0, 0.8]),
    ('QuasiPeriodic', [1.1, 30.0, 12.5, 0.6]),
    ('RationalQuadratic', [1.4, 0.8, 9.0]),
    ('RQP', [1.2, 0.9, 20.0, 13.0, 0.7]),
    ('Cosine', [0.8, 9.5]),
    ('Exponential', [1.1, 6.0]),
    ('Matern32', [1.3, 8.0]),
    ('Matern52', [0.7, 5.0]),
    ('GammaExp', [1.2, 1.5, 6.0]),
    ('Piecewise', [14.0]),
    ('Paciorek', [1.1, 5.0, 9.0]),
    ('NewPeriodic', [1.2, 0.9, 10.0, 0.8]),
    ('QuasiNewPeriodic', [1.1, 0.7, 25.0, 10.0, 0.9]),
    ('CosPeriodic', [1.3, 11.0, 0.9]),
    ('QuasiCosPeriodic', [0.9, 22.0, 9.0, 0.8]),
    ('Polynomial', [1.0, 0.01, 1.5, 2.0]),
    ('HarmonicPeriodic', [2, 1.1, 13.0, 0.9]),
    ('QuasiHarmonicPeriodic', [2, 1.2, 25.0, 11.0, 0.8]),
]
COMPOSITE_CASES = [
    # (tag, python expression over `c` = covfunc module)
    ('SE_plus_M32', 'c.SquaredExponential(1.1, 8.0) + c.Matern32(0.4, 3.0)'),
    ('SE_times_P', 'c.SquaredExponential(1.0, 10.0) * c.Periodic(1.0, 20.0, 0.5)'),
    ('sum_of_prod', 'c.SquaredExponential(0.9, 12.0) * c.Periodic(1.0, 7.0, 0.9) + c.Exponential(0.3, 4.0)'),
    ('dSE', 'c.Derivative(c.SquaredExponential(1.2, 6.0))'),
    ('dP', 'c.Derivative(c.Periodic(0.9, 11.0, 0.8))'),
    ('dQP', 'c.Derivative(c.QuasiPeriodic(1.1, 30.0, 12.5, 0.6))'),
]


def gen_kernels():
    rng = np.random.RandomState(11)
    t = np.sort(rng.uniform(0, 60, 24))
    dummy = rinference(1, t, np.zeros(24), np.ones(24))
    out = {'time': t}
    meta = {'simple': KERNEL_CASES, 'composite': COMPOSITE_CASES}
    for name, pars in KERNEL_CASES:
        k = getattr(rcov, name)(*pars)
        out['K_' + name] = dummy._KMatrix(k, t)
    for tag, expr in COMPOSITE_CASES:
        k = eval(expr, {'c': rcov})
        out['K_' + tag] = dummy._KMatrix(k, t)
    # rectangular evaluation (non-square r) pins WhiteNoise's other branch
    r = t[:5, None] - t[None, :]
    out['rect_r'] = r
    out['rect_WhiteNoise'] = rcov.WhiteNoise(0.7)(r)
    out['rect_QuasiPeriodic'] = rcov.QuasiPeriodic(1.1, 30.0, 12.5, 0.6)(r)
    np.savez_compressed(os.path.join(OUT, 'kernels.npz'), **out)
    with open(os.path.join(OUT, 'kernels.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print('kernels: %d matrices' % (len(out) - 1))


# --------------------------------------------------------- model problems
def model_spec(p, q, node_kind, nonzero_means):
    nodes, weights, means, jitters = synth.component_spec(p, q, node_kind)
    if nonzero_means:
        choices = [('Constant', [0.7]), ('Linear', [0.01, -0.4]),
                   ('Sine', [0.8, 17.0, 0.3]), None]
        means = [choices[i % len(choices)] for i in range(p)]
        jitters = [0.3 + 0.25 * i for i in range(p)]
    return nodes, weights, means, jitters


def make_ref(N, p, q, spec, seed=0):
    t, ys, es = synth.rv_series(N, p, seed)
    args = []
    for y, e in zip(ys, es):
        args += [y, e]
    g = rinference(q, t, *args)
    nodes, weights, means, jitters = synth.build_components(rcov, rmean, spec)
    g.set_components(nodes, weights, means, jitters)
    return g, t, ys, es


def setup_like_elbocalc(g):
    """The setup block of ELBOcalc, meanfield.py:618-624."""
    from gpyrn.meanfield import _cholNugget
    j2 = np.array(g.jitters) ** 2
    Kf = np.array([g._KMatrix(i, g.time) for i in g.nodes])
    Kw = np.array([g._KMatrix(j, g.time) for j in g.weights])
    Lf = np.array([_cholNugget(j)[0] for j in Kf])
    Lw = np.array([_cholNugget(j)[0] for j in Kw])
    y = np.concatenate(g.y) - g._mean(g.means)
    y = np.array(np.array_split(y, g.p))
    return Kf, Kw, Lf, Lw, y, j2


# Round 6 (missing #2): model problems on kernels other than SE / QP, and an ill-conditioned prior.
#   illc_*  : the shape of the one randomised problem that missed 1e-8 in round 5 (tests' seed 7: q = 3, p = 2, a pure
#             Periodic weight -- rank-deficient but for the reference's nugget -- among Sum / Multiplication / Matern / QP
#             kernels; the reference's Jacobi iteration diverges at q = 3, which takes the means far outside the range of
#             K, where m^T K^-1 m dominates the ELBO and carries eps cond(K)): forced sweeps at N = 100 (one tile) and 300
#   kmix_*  : a converging problem (q = 2) on Periodic, Multiplication, Matern52, RationalQuadratic, Sum, Matern32
# Composite kernels are written (name, [spec, spec]); synth.build_components builds them recursively.
ILLC_SPEC = ([('QuasiPeriodic', [1.2, 30.0, 22.0, 1.4]),
              ('Sum', [('SquaredExponential', [0.87, 29.5]), ('Exponential', [0.26, 14.7])]),
              ('SquaredExponential', [0.97, 8.8])],
             [('Multiplication', [('SquaredExponential', [0.53, 38.6]), ('Cosine', [1.0, 10.6])]),
              ('Matern32', [0.55, 28.4]),
              ('Sum', [('SquaredExponential', [1.39, 21.7]), ('Exponential', [0.42, 10.8])]),
              ('Periodic', [1.34, 22.7, 0.82]),
              ('Matern52', [1.33, 32.0]),
              ('Periodic', [0.58, 9.26, 1.47])],
             [('Linear', [0.006, -0.27]), ('Constant', [0.025])],
             [0.234, 0.229])
KMIX_SPEC = ([('Periodic', [1.1, 25.0, 0.9]),
              ('Multiplication', [('SquaredExponential', [0.9, 40.0]), ('Periodic', [1.0, 12.5, 0.8])])],
             [('Matern52', [0.8, 45.0]),
              ('RationalQuadratic', [1.0, 1.3, 60.0]),
              ('Sum', [('SquaredExponential', [0.7, 55.0]), ('Exponential', [0.2, 30.0])]),
              ('Matern32', [0.9, 70.0])],
             [('Constant', [0.3]), ('Linear', [0.004, -0.2])],
             [0.4, 0.55])


def irregular_series(N, p, seed, span):
    """(t, ys, yerrs): irregular sampling over `span` days, small error bars (the randomised tests' kind of data)."""
    rng = np.random.RandomState(seed)
    t = np.sort(rng.uniform(0.0, span, N))
    ys = [np.sin(2 * np.pi * t / rng.uniform(9.0, 30.0) + i) * rng.uniform(0.5, 2.0) + 0.02 * t * rng.randn()
          + 0.2 * rng.randn(N) for i in range(p)]
    es = [0.05 + 0.1 * rng.rand(N) for _ in range(p)]
    return t, ys, es


def gen_step_case(tag, N, p, q, node_kind, nonzero_means, nsweeps, full_calc,
                  keep_matrices=False, seed=0, spec=None, span=None):
    t0 = time.time()
    if spec is None:
        spec = model_spec(p, q, node_kind, nonzero_means)
    if span is None:
        g, t, ys, es = make_ref(N, p, q, spec, seed)
    else:
        t, ys, es = irregular_series(N, p, seed, span)
        g = rinference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(*synth.build_components(rcov, rmean, spec))
    out = {'time': t, 'y': np.array(ys), 'yerr': np.array(es)}
    meta = {'N': N, 'p': p, 'q': q, 'seed': seed,
            'nodes': spec[0], 'weights': spec[1], 'means': spec[2],
            'jitters': spec[3], 'nsweeps': nsweeps}

    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    out['mu_init'], out['var_init'] = mu0, var0
    f, w = g._u_to_fhatW(mu0)
    out['mu_init_f'], out['mu_init_w'] = f, w

    Kf, Kw, Lf, Lw, y, j2 = setup_like_elbocalc(g)
    out['y_resid'] = y
    out['logdiag_Lf'] = np.array([np.sum(np.log(np.diag(L))) for L in Lf])
    out['logdiag_Lw'] = np.array([np.sum(np.log(np.diag(L))) for L in Lw])
    if keep_matrices:
        out['Kf'], out['Kw'] = Kf, Kw

    # forced sweeps, each split into its parts (ELBOaux body, meanfield.py:682-710)
    mu, var = mu0, var0
    elbo, parts = [], []
    for s in range(nsweeps):
        muF, muW = g._u_to_fhatW(mu.flatten())
        varF, varW = g._u_to_fhatW(var.flatten())
        sigF, muFn, sigW, muWn = g._updateSigMu(Kf, Kw, Lf, Lw, y, j2,
                                                muF, varF, muW, varW)
        muFn3 = muFn.reshape(1, g.q, g.N)
        ent = float(g._entropy(sigF, sigW))
        logp = float(g._expectedLogPrior(Kf, Kw, Lf, Lw, sigF, muFn3, sigW, muWn))
        logl = float(g._expectedLogLike(y, j2, sigF, muFn3, sigW, muWn))
        E, mu, var, sF, sW = g.ELBOaux(Kf, Kw, Lf, Lw, y, j2, mu, var)
        assert abs(float(E) - (logl + logp + ent) / g.q) <= 1e-9 * abs(float(E))
        elbo.append(float(E))
        parts.append([logl, logp, ent])
        if s == 0 and keep_matrices:
            out['sigmaF_1'], out['sigmaW_1'] = sF, sW
        if s == 0:
            out['mu_1'], out['var_1'] = mu, var
    out['elbo_sweeps'] = np.array(elbo)
    out['parts_sweeps'] = np.array(parts)
    out['mu_final'], out['var_final'] = mu, var

    if full_calc:
        rec = []
        orig = g.ELBOaux

        def spy(*a, **k):
            r = orig(*a, **k)
            rec.append(float(r[0]))
            return r
        g.ELBOaux = spy
        try:
            E, muc, varc, it = g.ELBOcalc()
            out['calc_elbo'] = np.array(float(E))
            out['calc_mu'], out['calc_var'] = muc, varc
            out['calc_iter'] = np.array(it)
            out['calc_elbo_array'] = np.array(rec)
            # warm start, as nELBO does (meanfield.py:1102-1104)
            rec.clear()
            E2, mu2, var2, it2 = g.ELBOcalc(mu='previous', var='previous')
            out['warm_elbo'] = np.array(float(E2))
            out['warm_iter'] = np.array(it2)
            out['warm_elbo_array'] = np.array(rec)
        except np.linalg.LinAlgError:
            # NumPy raises where jax would return NaN: the
            # reference's explicitly formed Sigma lost positive-definiteness to
            # round-off, so there is no finite reference value to pin.
            meta['calc_failed_after'] = len(rec)
            out['calc_elbo_array_partial'] = np.array(rec)
            print('   %s: reference ELBOcalc hit a non-PD Sigma after %d ELBOaux calls'
                  % (tag, len(rec)))
        g.ELBOaux = orig
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **out)
    with open(os.path.join(OUT, tag + '.json'), 'w') as fjs:
        json.dump(meta, fjs, indent=1)
    print('%s: N=%d p=%d q=%d sweeps=%d ELBO[0]=%.12g  (%.1fs)'
          % (tag, N, p, q, nsweeps, elbo[0], time.time() - t0))


def gen_predict(tag, nstar=37):
    """inference._Prediction (meanfield.py:1289-1381) on the state after the forced sweeps of `tag`."""
    meta = json.load(open(os.path.join(OUT, tag + '.json')))
    d = np.load(os.path.join(OUT, tag + '.npz'))
    spec = (meta['nodes'], meta['weights'], meta['means'], meta['jitters'])
    g, t, ys, es = make_ref(meta['N'], meta['p'], meta['q'], spec, meta['seed'])
    span = t.max() - t.min()
    tstar = np.linspace(t.min() - 0.2 * span, t.max() + 0.2 * span, nstar)
    mean, var, sep = g._Prediction(tstar=tstar, mu=d['mu_final'], var=d['var_final'], separate=True)
    np.savez_compressed(os.path.join(OUT, 'pred_' + tag + '.npz'), tstar=tstar, mean=mean, var=var,
                        node_means=np.array(sep[0], dtype=float), weight_means=np.array(sep[1], dtype=float))
    print('pred_%s: %d points, mean[0]=%s' % (tag, nstar, mean[0]))


def gen_api():
    """Parameter plumbing contracts, meanfield.py:180-379."""
    spec = model_spec(2, 2, 'QP', True)
    spec = (spec[0], spec[1], [('Constant', [0.7]), ('Linear', [0.01, -0.4])], spec[3])
    g, *_ = make_ref(16, 2, 2, spec)
    out = {}
    out['names'] = list(g.parameters_dict.keys())
    out['values'] = [float(v) for v in g.parameters_dict.values()]
    out['get_all'] = g.get_parameters(include_frozen=True).tolist()
    out['n_parameters'] = int(g.n_parameters)
    g.freeze_parameter(name='node1*')
    g.freeze_parameter(index=9)
    out['mask_after_freeze'] = g.frozen_mask.tolist()
    out['get_free'] = g.get_parameters().tolist()
    g.thaw_parameter(name='node1.P')
    out['mask_after_thaw'] = g.frozen_mask.tolist()
    newp = np.arange(1, g.n_parameters + 1, dtype=float) / 10
    g.set_parameters(newp.copy())
    out['after_set_all'] = g.get_parameters(include_frozen=True).tolist()
    free = g.get_parameters()
    g.set_parameters(free * 2)
    out['after_set_free'] = g.get_parameters(include_frozen=True).tolist()
    out['jitters_after'] = np.asarray(g.jitters).tolist()
    with open(os.path.join(OUT, 'api.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('api: %d parameters' % out['n_parameters'])


def gen_traj_case(tag, N, p, q, node_kind, seed=0):
    """Full ELBOcalc trajectory at a BASELINE config (meanfield.py:561-649): every ELBOaux value the loop
    sees, iterNumber and the converged mu/var, then the warm start nELBO uses (meanfield.py:1102-1104)."""
    t0 = time.time()
    spec = model_spec(p, q, node_kind, False)
    g, t, ys, es = make_ref(N, p, q, spec, seed)
    rec = []
    orig = g.ELBOaux

    def spy(*a, **k):
        r = orig(*a, **k)
        rec.append(float(r[0]))
        print('   %s: ELBOaux #%d = %.15g (%.0fs)' % (tag, len(rec), rec[-1], time.time() - t0), flush=True)
        return r
    g.ELBOaux = spy
    out = {}
    E, muc, varc, it = g.ELBOcalc()
    out['calc_elbo'] = np.array(float(E))
    out['calc_mu'], out['calc_var'] = muc, varc
    out['calc_iter'] = np.array(it)
    out['calc_elbo_array'] = np.array(rec)
    rec.clear()
    E2, mu2, var2, it2 = g.ELBOcalc(mu='previous', var='previous')
    out['warm_elbo'] = np.array(float(E2))
    out['warm_iter'] = np.array(it2)
    out['warm_elbo_array'] = np.array(rec)
    out['warm_mu'], out['warm_var'] = mu2, var2
    g.ELBOaux = orig
    meta = {'N': N, 'p': p, 'q': q, 'seed': seed, 'nodes': spec[0], 'weights': spec[1],
            'means': spec[2], 'jitters': spec[3]}
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **out)
    with open(os.path.join(OUT, tag + '.json'), 'w') as fjs:
        json.dump(meta, fjs, indent=1)
    print('%s: N=%d p=%d q=%d calc_iter=%d ELBO=%.15g warm_iter=%d (%.1fs)'
          % (tag, N, p, q, it, float(E), it2, time.time() - t0), flush=True)


def gen_big_sweep(tag, N, seed=0):
    """One reference-form sweep (ELBOaux body, meanfield.py:682-710) at a size beyond the other fixtures,
    p = q = 1: scalars and the O(N) state only -- no matrix is stored."""
    t0 = time.time()
    spec = model_spec(1, 1, 'QP', False)
    g, t, ys, es = make_ref(N, 1, 1, spec, seed)
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    Kf, Kw, Lf, Lw, y, j2 = setup_like_elbocalc(g)
    print('   %s: setup done (%.0fs)' % (tag, time.time() - t0), flush=True)
    out = {'logdiag_Lf': np.array([np.sum(np.log(np.diag(L))) for L in Lf]),
           'logdiag_Lw': np.array([np.sum(np.log(np.diag(L))) for L in Lw])}
    muF, muW = g._u_to_fhatW(mu0.flatten())
    varF, varW = g._u_to_fhatW(var0.flatten())
    sigF, muFn, sigW, muWn = g._updateSigMu(Kf, Kw, Lf, Lw, y, j2, muF, varF, muW, varW)
    print('   %s: _updateSigMu done (%.0fs)' % (tag, time.time() - t0), flush=True)
    muFn3 = muFn.reshape(1, g.q, g.N)
    ent = float(g._entropy(sigF, sigW))
    logp = float(g._expectedLogPrior(Kf, Kw, Lf, Lw, sigF, muFn3, sigW, muWn))
    logl = float(g._expectedLogLike(y, j2, sigF, muFn3, sigW, muWn))
    varFn = np.array([np.diag(s) for s in sigF])
    varWn = np.array([[np.diag(s) for s in row] for row in sigW])
    out['elbo_sweeps'] = np.array([(logl + logp + ent) / g.q])
    out['parts_sweeps'] = np.array([[logl, logp, ent]])
    out['mu_f_1'], out['mu_w_1'] = np.asarray(muFn), np.asarray(muWn)
    out['var_f_1'], out['var_w_1'] = varFn, varWn
    meta = {'N': N, 'p': 1, 'q': 1, 'seed': seed, 'nodes': spec[0], 'weights': spec[1],
            'means': spec[2], 'jitters': spec[3], 'nsweeps': 1}
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **out)
    with open(os.path.join(OUT, tag + '.json'), 'w') as fjs:
        json.dump(meta, fjs, indent=1)
    print('%s: N=%d ELBO=%.15g parts=%s (%.1fs)' % (tag, N, out['elbo_sweeps'][0], out['parts_sweeps'][0],
                                                   time.time() - t0), flush=True)


def gen_optimize(tag='opt_N64_p2q1'):
    """inference.optimize (meanfield.py:1114-1152): a short deterministic Nelder-Mead run."""
    N, p, q = 64, 2, 1
    spec = model_spec(p, q, 'QP', True)
    spec = (spec[0], spec[1], [('Constant', [0.7]), ('Linear', [0.01, -0.4])], spec[3])
    g, t, ys, es = make_ref(N, p, q, spec)
    x0 = g.get_parameters().copy()
    calls = []
    orig = g.nELBO

    def spy(x, *a, **k):
        v = orig(x, *a, **k)
        calls.append([float(v)] + [float(xx) for xx in x])
        return v
    g.nELBO = spy
    res = g.optimize(options={'maxiter': 10})
    out = {'x0': x0.tolist(), 'x': res.x.tolist(), 'fun': float(res.fun), 'nfev': int(res.nfev),
           'nit': int(res.nit), 'calls': calls, 'names': list(g.parameters_dict.keys()),
           'N': N, 'p': p, 'q': q, 'nodes': spec[0], 'weights': spec[1], 'means': spec[2], 'jitters': spec[3]}
    # vars= forms: only one parameter free / all but one
    g2, *_ = make_ref(N, p, q, spec)
    r2 = g2.optimize(vars='node1.P', options={'maxiter': 6})
    out['vars_P'] = {'x': r2.x.tolist(), 'fun': float(r2.fun), 'mask': g2.frozen_mask.tolist(),
                     'all': g2.get_parameters(include_frozen=True).tolist()}
    with open(os.path.join(OUT, tag + '.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('%s: fun=%.12g nfev=%d' % (tag, out['fun'], out['nfev']))


def gen_multiconstant(tag='multiconstant'):
    """meanfunc.MultiConstant (meanfunc.py:138-187) alone and inside ELBOcalc via inference._mean."""
    rng = np.random.RandomState(5)
    N = 48
    t = np.sort(rng.uniform(0, 90, N))
    obsid = np.r_[np.full(17, 1), np.full(12, 2), np.full(19, 3)].astype(float)
    m = rmean.MultiConstant([1.5, -0.75, 3.25], obsid, t)
    tq = np.linspace(-5, 95, 31)
    out = {'time': t, 'obsid': obsid, 'offsets': np.array([1.5, -0.75, 3.25]), 'at_time': m(t),
           'tq': tq, 'at_tq': m(tq), 'time_bins': m.time_bins(), 'parsize': np.array(m._parsize)}
    names = list(m._param_names)
    # inside the model: one output with instrument offsets, one node, SE kernels
    y = 2.0 * np.sin(2 * np.pi * t / 20) + np.take([1.5, -0.75, 0.0], obsid.astype(int) - 1) + 3.25 \
        + rng.normal(0, 0.3, N)
    e = rng.uniform(0.2, 0.4, N)
    g = rinference(1, t, y, e)
    g.set_components(rcov.SquaredExponential(1.1, 15.0), rcov.SquaredExponential(0.9, 40.0),
                     rmean.MultiConstant([1.5, -0.75, 3.25], obsid, t), 0.4)
    E, mu, var, it = g.ELBOcalc()
    out.update(y=y, yerr=e, calc_elbo=np.array(float(E)), calc_mu=mu, calc_var=var, calc_iter=np.array(it),
               mean_vec=g._mean(g.means))
    out['pnames'] = np.array(list(g.parameters_dict.keys()))
    out['pvalues'] = np.array(list(g.parameters_dict.values()), dtype=float)
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **out)
    print('%s: names=%s ELBO=%.12g iter=%d' % (tag, names, float(E), it))


def gen_mcmc(tag='mcmc_N48_p1q1'):
    """inference.mcmc (meanfield.py:1154-1286) under the deterministic emcee stand-in of tests/fake_emcee
    (emcee itself is not installed anywhere here): the reference's chain, log-probabilities and ELBO blobs
    for a seeded run.  Run as `python oracle/gen_golden.py mcmc` -- it needs the stand-in in front of
    oracle/standins, which this function arranges by re-importing."""
    import importlib
    import scipy.stats as st
    fake = os.path.join(REPO, 'tests', 'fake_emcee')
    for name in [m for m in sys.modules if m == 'emcee' or m.startswith('emcee.')]:
        del sys.modules[name]
    sys.path.insert(0, fake)
    import emcee
    assert emcee.__version__.endswith('fake')
    import gpyrn.meanfield as rm
    rm.EnsembleSampler = emcee.EnsembleSampler          # the names meanfield.py imported at load time
    rm.backends = emcee.backends
    rm.sample_ellipsoid = emcee.utils.sample_ellipsoid
    N, p, q = 48, 1, 1
    spec = ([('QuasiPeriodic', [1.1, 40.0, 23.0, 0.8])], [('SquaredExponential', [0.9, 55.0])],
            [('Constant', [0.2])], [0.45])
    out = {'N': N, 'p': p, 'q': q, 'nodes': spec[0], 'weights': spec[1], 'means': spec[2], 'jitters': spec[3],
           'seed': 2024, 'niter': 12, 'vars': ['node1.P', 'weight1.ell', 'jitter1']}

    def priors():
        return {'node1.P': st.uniform(15.0, 20.0), 'weight1.ell': st.uniform(30.0, 60.0),
                'jitter1': st.uniform(0.05, 1.5)}
    for label, p0 in (('prior_start', None), ('ellipsoid_start', [23.0, 55.0, 0.45])):
        g, *_ = make_ref(N, p, q, spec)
        np.random.seed(out['seed'])
        sampler = g.mcmc(priors(), p0=p0, vars=list(out['vars']), niter=out['niter'])
        out[label] = {'chain': sampler.get_chain().tolist(), 'log_prob': sampler.get_log_prob().tolist(),
                      'blobs': sampler.get_blobs().tolist(), 'iteration': int(sampler.iteration),
                      'tau': sampler.get_autocorr_time(tol=0).tolist(),
                      'final_parameters': g.get_parameters(include_frozen=True).tolist(),
                      'mask': g.frozen_mask.tolist()}
    with open(os.path.join(OUT, tag + '.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('%s: %d iterations, last log-prob max %.10g' % (tag, out['prior_start']['iteration'],
                                                        max(out['prior_start']['log_prob'][-1])))


def gen_solar(tag='solar'):
    """The reference's data file gpyrn/datasets/Solar_observations.txt (header line 1, 13 columns): per-column
    sums and the first/last rows, for the loader test.  The values are data, read with numpy alone."""
    path = os.path.join(REF, 'gpyrn', 'datasets', 'Solar_observations.txt')
    with open(path) as f:
        header = f.readline().split()
    d = np.loadtxt(path, skiprows=1)
    out = {'header': header, 'shape': list(d.shape), 'colsum': d.sum(axis=0).tolist(),
           'first': d[0].tolist(), 'last': d[-1].tolist(), 'bytes': os.path.getsize(path)}
    with open(os.path.join(OUT, tag + '.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('%s: shape %s header %s' % (tag, d.shape, header))


if __name__ == '__main__':
    big = '--big' in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith('--')]

    def want(tag):
        return not only or tag in only

    if want('kernels'):
        gen_kernels()
    if want('api'):
        gen_api()
    cases = [
        # tag, N, p, q, node, nonzero means, forced sweeps, full ELBOcalc, keep matrices
        ('step_p1q1', 32, 1, 1, 'SE', True, 3, True, True),
        ('step_p2q1', 32, 2, 1, 'QP', True, 3, True, False),
        ('step_p1q2', 32, 1, 2, 'QP', True, 3, True, False),
        ('step_p3q2', 32, 3, 2, 'QP', True, 3, True, True),
        ('step_p2q3', 40, 2, 3, 'QP', True, 3, True, False),
        ('cfg1_N200', 200, 1, 1, 'SE', False, 6, True, False),
        ('mid_N300_p3q2', 300, 3, 2, 'QP', True, 4, True, False),
        ('mid_N512_p3q2', 512, 3, 2, 'QP', False, 3, False, False),
        ('mid_N1024_p1q1', 1024, 1, 1, 'QP', False, 3, False, False),
    ]
    if big:
        cases += [
            ('cfg2_N2048', 2048, 1, 1, 'QP', False, 3, False, False),
            ('cfg3_N4096', 4096, 3, 2, 'QP', False, 2, False, False),
            # BASELINE config 4 (16 latent GPs, four nodes: the cumulative-trace quirk at full size)
            ('cfg4_N4096_q4', 4096, 3, 4, 'QP', False, 1, False, False),
            # BASELINE config 5's SHAPE (p = 4, q = 3: the cumulative-trace quirk Q1 with three nodes and the
            # raw-reshape quirk Q2 with four outputs) at sizes the reference finishes in minutes
            ('cfg5shape_N1024', 1024, 4, 3, 'QP', False, 2, False, False),
            ('cfg5shape_N2048', 2048, 4, 3, 'QP', False, 2, False, False),
        ]
    for c in cases:
        if want(c[0]):
            gen_step_case(*c)
    # (tag, N, p, q, forced sweeps, full ELBOcalc, spec, span of the sampling)
    for tag, N, p, q, ns, full, spec, span in (('illc_N100_p2q3', 100, 2, 3, 3, False, ILLC_SPEC, 80.0),
                                               ('illc_N300_p2q3', 300, 2, 3, 3, False, ILLC_SPEC, 240.0),
                                               # (eight tiles: the launch schedule's outer panels, K = 512 updates; cond(K) 4e8)
                                               ('illc_N1000_p2q3', 1000, 2, 3, 2, False, ILLC_SPEC, 800.0),
                                               ('kmix_N200_p2q2', 200, 2, 2, 4, True, KMIX_SPEC, 160.0)):
        if want(tag):
            gen_step_case(tag, N, p, q, None, True, ns, full, False, 7, spec, span)
    for tag in ('step_p1q1', 'step_p3q2', 'step_p2q3', 'cfg1_N200', 'mid_N300_p3q2'):
        if want('pred_' + tag):
            gen_predict(tag)
    if want('optimize'):
        gen_optimize()
    if want('multiconstant'):
        gen_multiconstant()
    if want('solar'):
        gen_solar()
    if 'mcmc' in only:                  # swaps the emcee stand-in: only on request, and last
        gen_mcmc()
    if big:
        # trajectories at the BASELINE configs (cfg 3: ~21 reference sweeps of ~35 s) and one reference-form
        # sweep each at N = 8192 and N = 16384 (7.2.N^3 flop: minutes to tens of minutes, ~30 GB)
        for tag, N, p, q in (('traj_cfg2_N2048', 2048, 1, 1), ('traj_cfg3_N4096', 4096, 3, 2)):
            if want(tag):
                gen_traj_case(tag, N, p, q, 'QP')
        for tag, N in (('big_N8192', 8192), ('big_N16384', 16384)):
            if want(tag):
                gen_big_sweep(tag, N)
