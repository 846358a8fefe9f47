"""Parity of the HIP path with the reference: golden vectors generated from the
reference itself (tests/golden, oracle/gen_golden.py) and the CPU oracle on
seeded inputs.  Everything goes through gpyrn_amd's public API and therefore the
C ABI.  Tolerance: 1e-8 relative on the ELBO and posterior means (north_star)."""
import json
import os

import numpy as np
import pytest

import gpyrn_amd as gpyrn
from gpyrn_amd import _hip, covfunc, meanfunc, synth
from oracle import cpu_ref
from tests import _cases

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def _model(tag):
    meta, d = _cases.load(tag)
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    g = gpyrn.inference(meta['q'], np.array(d['time']), *_cases.data_args(d))
    g.set_components(nodes, weights, means, jit)
    return meta, d, g


# ---------------------------------------------------------------- covariance fill
def test_fill_matches_reference_kernels():
    with open(os.path.join(_cases.GOLDEN, 'kernels.json')) as f:
        meta = json.load(f)
    d = np.load(os.path.join(_cases.GOLDEN, 'kernels.npz'))
    t = d['time']
    g = gpyrn.inference(1, t, np.zeros(t.size), np.ones(t.size))
    for name, pars in meta['simple']:
        k = getattr(covfunc, name)(*pars)
        assert k._device_program() is not None, name
        K = g._KMatrix(k)
        np.testing.assert_allclose(K, d['K_' + name], rtol=1e-12, atol=1e-13, err_msg=name)
    for tag, expr in meta['composite']:
        k = eval(expr, {'c': covfunc})
        assert k._device_program() is not None, tag
        K = g._KMatrix(k)
        np.testing.assert_allclose(K, d['K_' + tag], rtol=1e-12, atol=1e-12, err_msg=tag)


@pytest.mark.parametrize('n', [1, 2, 63, 65, 129, 257])
def test_fill_sizes_around_the_block_edges(n):
    """k_fill_sym evaluates two adjacent columns per thread and mirrors 64 x 64 blocks: sizes that end inside a pair, a block
    and a tile, against the host formulas (gpyrn_amd.covfunc, themselves checked against the reference's matrices in
    test_api.py) -- SE / Periodic / QP go through exp_neg, div_rn and sin_sq_rad on the device."""
    rng = np.random.default_rng(n)
    t = np.sort(rng.uniform(0.0, 400.0, n))
    g = gpyrn.inference(1, t, np.zeros(n), np.ones(n))
    r = t[:, None] - t[None, :]
    for k in (covfunc.SquaredExponential(1.3, 7.0), covfunc.Periodic(0.8, 23.0, 0.9),
              covfunc.QuasiPeriodic(1.1, 31.0, 23.0, 0.7), covfunc.Matern32(0.9, 12.0),
              covfunc.QuasiPeriodic(1.1, 31.0, 23.0, 0.7) + covfunc.SquaredExponential(0.3, 2.0)):
        assert k._device_program() is not None
        K = g._KMatrix(k)
        want = k(r) + 1e-6 * np.eye(n)          # meanfield.py:433: the nugget of _KMatrix
        np.testing.assert_allclose(K, want, rtol=1e-12, atol=1e-14, err_msg=type(k).__name__)
        assert np.array_equal(K, K.T)


def test_fill_keeps_the_references_rounding_sequence():
    """Round 6: on a prior with cond(K) ~ 1e9 the LAST BITS of K's entries are worth 1e-8 on m^T K^-1 m
    (profiles/r06_fill_rounding.txt).  The reference writes np.sin(np.pi * np.abs(r) / P)**2 and x / ell**2: the sine of a
    ROUNDED argument hundreds of periods out, IEEE quotients.  The fill follows that sequence (csrc/fill.hip: div_rn,
    sin_sq_rad), so its matrices differ from NumPy's evaluation of the same formula (gpyrn_amd.covfunc's __call__, pinned to
    the reference's matrices by kernels.npz) by the last bit or two of exp / sin -- a fraction of an ulp on average, where
    rounds 1-5 (sin^2 of the exact fraction, products with reciprocals) were 13-42 ulp away on average and 500-1000 at worst."""
    rng = np.random.RandomState(3)
    n = 1000
    t = np.sort(rng.uniform(0.0, 800.0, n))
    g = gpyrn.inference(1, t, np.zeros(n), np.ones(n))
    r = t[:, None] - t[None, :]
    for k, mean_bound in ((covfunc.SquaredExponential(0.97, 8.8), 0.25), (covfunc.Periodic(1.34, 22.7, 0.82), 1.5),
                          (covfunc.QuasiPeriodic(1.2, 30.0, 22.0, 1.4), 1.0)):
        K = g._KMatrix(k)
        want = k(r) + 1e-6 * np.eye(n)
        big = want > 1e-200                                   # (entries in the denormal tail of the exponential aside)
        ulp = np.abs(K - want)[big] / np.spacing(want[big])
        assert ulp.mean() <= mean_bound, '%s: %.2f ulp on average' % (type(k).__name__, ulp.mean())
        # (exp(x) has condition number |x|: where the last bit of sin**2 flips the rounding of term1 - term2, an entry
        # exp(-150) jumps by 150 / 2 ulp -- whichever libm computed the sine.  The entries that carry weight in a
        # factorisation, those within 1e-6 of the largest, have |x| < 14.)
        rel = want[big] > 1e-6 * want.max()
        assert np.percentile(ulp[rel], 99.9) <= 16, '%s: 99.9th percentile %.0f ulp' % (type(k).__name__, np.percentile(ulp[rel], 99.9))


def test_user_kernel_takes_host_path():
    class MySE(covfunc.covFunction):
        _param_names = ('a', 'l')

        def __call__(self, r):
            return self.pars[0]**2 * np.exp(-0.5 * r**2 / self.pars[1]**2)

    meta, d, g = _model('step_p1q1')
    e_builtin = g.ELBOcalc()[0]
    nodes = [MySE(*n.pars) for n in g.nodes]
    assert nodes[0]._device_program() is None
    g.set_components(nodes, g.weights, g.means, g.jitters)
    e_user = g.ELBOcalc()[0]
    np.testing.assert_allclose(e_user, e_builtin, rtol=1e-10)


@pytest.mark.parametrize('tag', ['mid_N300_p3q2', 'cfg1_N200'])
def test_user_kernels_above_one_tile_mix_with_device_kernels(tag):
    """User-defined covFunction subclasses (host-evaluated, gprn_upload_K) on the LAUNCH path (N > 128), mixed with device
    programs in one problem: the second node (the one whose K_j^-1 quirk Q1 needs when q = 2) and every other weight are
    user kernels that evaluate the fixture's own formula on the host.  The whole ELBOcalc -- the unsharded set-up
    (factor_priors_single: one factorisation over uploaded and filled matrices alike), trip count, value, state -- must
    reproduce the all-device run, which reproduces the reference."""
    if not _cases.available(tag):
        pytest.skip('fixture not generated')

    class Hosted(covfunc.covFunction):
        def __init__(self, inner):
            super().__init__(*inner.pars)
            self.inner = inner

        def __call__(self, r):
            return self.inner(r)

    meta, d, g = _model(tag)
    e_dev, mu_dev, var_dev, it_dev = g.ELBOcalc()
    nodes = [Hosted(k) if j == len(g.nodes) - 1 else k for j, k in enumerate(g.nodes)]
    weights = [Hosted(k) if i % 2 == 0 else k for i, k in enumerate(g.weights)]
    assert nodes[-1]._device_program() is None
    g2 = _model(tag)[2]
    g2.set_components(nodes, weights, g2.means, g2.jitters)
    e_usr, mu_usr, var_usr, it_usr = g2.ELBOcalc()
    assert it_usr == it_dev and g2.last_info == 0
    np.testing.assert_allclose(e_usr, e_dev, rtol=1e-10)
    _cases.assert_state('user kernels ' + tag, mu_usr, mu_dev, var_usr, var_dev)
    _assert_default_schedule(g2._backend())


# ------------------------------------------------------------------- one sweep
def _assert_default_schedule(ctx):
    """No call of this context was re-run on HIP events after an in-kernel wait timed out, and it is still on the
    schedule it started with (at T = 128 the tests used to pass on the silent fallback).  Contexts that
    never had device-side flags (a serialising tool, GPRN_FLAGS=0) have nothing to fall back from."""
    assert ctx.option('fallbacks') == 0
    if os.environ.get('GPRN_FLAGS', '1') != '0' and not os.environ.get('ROCPROF_COUNTER_COLLECTION'):
        assert ctx.option('flags') == 1


# illc_*: an ill-conditioned prior under a diverging state (a pure Periodic weight, q = 3: where the explicit-inverse panel
# steps of rounds 1-5 missed 1e-8); kmix_*: a converging problem on Periodic / Multiplication / Matern / RationalQuadratic /
# Sum kernels -- both generated from the reference in round 6 (oracle/gen_golden.py)
SMALL = ['step_p1q1', 'step_p2q1', 'step_p1q2', 'step_p3q2', 'step_p2q3', 'illc_N100_p2q3']
MID = ['cfg1_N200', 'mid_N300_p3q2', 'mid_N512_p3q2', 'mid_N1024_p1q1', 'illc_N300_p2q3', 'kmix_N200_p2q2',
       'illc_N1000_p2q3']


@pytest.mark.parametrize('tag', SMALL + MID + ['cfg2_N2048', 'cfg3_N4096', 'cfg4_N4096_q4',
                                              'cfg5shape_N1024', 'cfg5shape_N2048'])
def test_forced_sweeps_match_reference(tag):
    if not _cases.available(tag):
        pytest.skip('fixture not generated')
    meta, d, g = _model(tag)
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    assert np.array_equal(mu0, d['mu_init']) and np.array_equal(var0, d['var_init'])
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    assert g.last_info == 0
    ctx.set_muvar(mu0, var0)
    elbo, parts, info = ctx.sweep(meta['nsweeps'], commit=True)
    assert info == 0
    np.testing.assert_allclose(elbo, d['elbo_sweeps'], rtol=RTOL)
    np.testing.assert_allclose(parts, d['parts_sweeps'], rtol=RTOL)
    mu, var = ctx.get_muvar()
    _cases.assert_state('forced sweeps ' + tag, mu, d['mu_final'], var, d['var_final'])      # north_star: 1e-8, norm-wise
    np.testing.assert_allclose(mu, d['mu_final'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var, d['var_final'], rtol=1e-6, atol=1e-12)
    # log det K of the setup (reference: sum log diag chol K)
    ld = ctx.get_logdet_K()
    np.testing.assert_allclose(ld[:meta['q']], 2 * d['logdiag_Lf'], rtol=1e-9)
    np.testing.assert_allclose(ld[meta['q']:], 2 * d['logdiag_Lw'], rtol=1e-9)


def test_first_sweep_state_and_uncommitted_sweep():
    meta, d, g = _model('step_p3q2')
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    ctx.set_muvar(d['mu_init'], d['var_init'])
    e0, _, _ = ctx.sweep(1, commit=False)
    mu, var = ctx.get_muvar()
    assert np.array_equal(mu.ravel(), d['mu_init'])            # untouched
    e1, _, _ = ctx.sweep(1, commit=True)
    assert e0[0] == e1[0]                                      # quirk Q7
    mu, var = ctx.get_muvar()
    _cases.assert_state('first sweep step_p3q2', mu, d['mu_1'], var, d['var_1'])
    np.testing.assert_allclose(mu, d['mu_1'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, d['var_1'], rtol=1e-6, atol=1e-12)


# ------------------------------------------------------------------- ELBOcalc
@pytest.mark.parametrize('tag', ['step_p1q1', 'step_p2q1', 'step_p1q2', 'step_p3q2',
                                 'cfg1_N200', 'mid_N300_p3q2', 'kmix_N200_p2q2'])
def test_elbocalc_trajectory(tag):
    meta, d, g = _model(tag)
    if 'calc_elbo' not in d:
        pytest.skip('reference produced no finite value')
    E, mu, var, it = g.ELBOcalc()
    assert it == int(d['calc_iter'])
    np.testing.assert_allclose(g._elbo_history, d['calc_elbo_array'], rtol=RTOL)
    np.testing.assert_allclose(E, float(d['calc_elbo']), rtol=RTOL)
    assert mu.shape == (meta['p'] + 1, meta['q'], meta['N'])
    _cases.assert_state('ELBOcalc ' + tag, mu, d['calc_mu'], var, d['calc_var'])
    np.testing.assert_allclose(mu, d['calc_mu'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var, d['calc_var'], rtol=1e-6, atol=1e-12)
    # warm start as nELBO does it (meanfield.py:1102-1104)
    E2, _, _, it2 = g.ELBOcalc(mu='previous', var='previous')
    assert it2 == int(d['warm_iter'])
    np.testing.assert_allclose(E2, float(d['warm_elbo']), rtol=RTOL)


def test_reference_own_inference_tests():
    """tests/test_inference.py:39-53 of the reference: bare objects or lists, ELBO runs."""
    rng = np.random.RandomState(0)
    t, y, yerr = rng.rand(3, 10)
    g = gpyrn.inference(1, t, y, yerr)
    node, weight = covfunc.SquaredExponential(1, 1), covfunc.SquaredExponential(1, 1)
    mean = meanfunc.Constant(0)
    g.set_components(node, weight, mean, 0.0)
    assert g.nodes[0] is node
    g.set_components([node], [weight], mean, 0.0)
    g.set_components([node], [weight], [mean], [0.0])
    e = g.ELBO
    # same inputs through the oracle
    Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, [node], [weight], [mean], [0.0], y[None])
    mu0, var0 = cpu_ref.init_mu_var(y[None], [1.0], [1.0], [0.0])
    e_ref = cpu_ref.elbo_calc(Kf, Kw, Lf, Lw, yres, y[None], yerr[None]**2, j2, mu0, var0)[0]
    np.testing.assert_allclose(e, e_ref, rtol=RTOL)


@pytest.mark.parametrize('n,p,q', [(1, 1, 1), (2, 1, 1), (3, 2, 1), (5, 2, 2), (17, 1, 2), (127, 1, 1), (129, 2, 1), (257, 1, 1)])
def test_elbocalc_at_sizes_around_the_tile_edges(n, p, q):
    """Whole ELBOcalc at sizes that leave the padded tile almost empty, fill a block exactly but for one row, or start a
    second tile with a single row (the factorisation runs on whole 128 x 128 tiles with identity padding; the fill works
    in column pairs and 64 x 64 blocks): against the CPU oracle on the same seeded inputs -- "parity unpinned" by the
    reference itself at these sizes, pinned through the oracle, which the golden vectors pin (test_oracle.py)."""
    rng = np.random.RandomState(100 * n + 10 * p + q)
    t = np.sort(rng.uniform(0.0, 60.0, n))
    ys = [np.sin(0.3 * t + i) + 0.1 * rng.randn(n) for i in range(p)]
    es = [0.05 + 0.05 * rng.rand(n) for _ in range(p)]
    nodes = [covfunc.QuasiPeriodic(1.0, 20.0 + j, 11.0, 0.8) for j in range(q)]
    weights = [covfunc.SquaredExponential(0.9 + 0.05 * k, 15.0 + k) for k in range(q * p)]   # node-major: j * p + i
    means = [meanfunc.Constant(0.1 * i) for i in range(p)]
    jit = [0.01] * p
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    elbo, mu, var, it = g.ELBOcalc()
    assert g.last_info == 0
    y = np.array(ys)
    Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, g.nodes, g.weights, g.means, g.jitters, y)
    mu0, var0 = cpu_ref.init_mu_var(y, [k.pars[0] for k in g.nodes], [k.pars[0] for k in g.weights], g.jitters)
    e_ref, mu_ref, var_ref, it_ref, _ = cpu_ref.elbo_calc(Kf, Kw, Lf, Lw, yres, y, np.array(es)**2, j2, mu0, var0, form='ref')
    assert it == it_ref
    np.testing.assert_allclose(elbo, e_ref, rtol=RTOL)
    _cases.assert_state('ELBOcalc N=%d p=%d q=%d vs oracle' % (n, p, q), mu, mu_ref, var, var_ref)
    np.testing.assert_allclose(mu, mu_ref, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var, var_ref, rtol=1e-6, atol=1e-10)


def _random_problem(seed):
    """A seeded problem no fixture holds: size, p, q, kernel families (built-ins with their own device programs, a product
    and a sum among them), mean functions, jitters, irregular sampling."""
    rng = np.random.RandomState(seed)
    n = int(rng.choice([3, 9, 31, 64, 100, 128, 129, 150, 200, 257, 300]))
    p, q = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    t = np.sort(rng.uniform(0.0, 80.0, n))
    ys = [np.sin(2 * np.pi * t / rng.uniform(9.0, 30.0) + i) * rng.uniform(0.5, 2.0) + 0.02 * t * rng.randn()
          + 0.2 * rng.randn(n) for i in range(p)]
    es = [0.05 + 0.1 * rng.rand(n) for _ in range(p)]

    def kernel(amp):
        kind = rng.randint(8)
        ell, per = rng.uniform(5.0, 40.0), rng.uniform(8.0, 25.0)
        if kind == 0:
            return covfunc.SquaredExponential(amp, ell)
        if kind == 1:
            return covfunc.Periodic(amp, per, rng.uniform(0.5, 1.5))
        if kind == 2:
            return covfunc.QuasiPeriodic(amp, ell, per, rng.uniform(0.5, 1.5))
        if kind == 3:
            return covfunc.RationalQuadratic(amp, rng.uniform(0.5, 3.0), ell)
        if kind == 4:
            return covfunc.Matern32(amp, ell)
        if kind == 5:
            return covfunc.Matern52(amp, ell)
        if kind == 6:
            return covfunc.SquaredExponential(amp, ell) * covfunc.Cosine(1.0, per)
        return covfunc.SquaredExponential(amp, ell) + covfunc.Exponential(0.3 * amp, 0.5 * ell)

    nodes = [kernel(rng.uniform(0.7, 1.3)) for _ in range(q)]
    weights = [kernel(rng.uniform(0.5, 1.5)) for _ in range(q * p)]
    means = []
    for i in range(p):
        kind = rng.randint(4)
        means.append(None if kind == 0 else meanfunc.Constant(rng.uniform(-0.5, 0.5)) if kind == 1 else
                     meanfunc.Linear(rng.uniform(-0.01, 0.01), rng.uniform(-0.3, 0.3)) if kind == 2 else
                     meanfunc.Sine(rng.uniform(0.1, 0.5), rng.uniform(10.0, 30.0), rng.uniform(0.0, 1.0)))
    jit = [float(rng.uniform(0.0, 0.3)) for _ in range(p)]
    return t, ys, es, nodes, weights, means, jit, p, q


@pytest.mark.parametrize('seed', range(14))
def test_randomised_problems_against_the_oracle(seed):
    """Seeded problems that no fixture holds (sizes from 3 to 300 either side of the tile edges, p and q up to 3, eight
    kernel families incl. a product and a sum, absent / constant / linear / sine means, zero to large jitters) through the
    whole ELBOcalc -- capped at eight trips, three at q = 3: the reference's Jacobi iteration need not converge -- against
    the CPU oracle's reference formulation on the same inputs: trip count, ELBO to 1e-8, state norm-wise 1e-8.  "Parity
    unpinned" by the reference itself on these inputs; pinned through the oracle, which the golden vectors pin."""
    t, ys, es, nodes, weights, means, jit, p, q = _random_problem(seed)
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    # (q = 3: the reference's iteration diverges, |ELBO| x 12-16 per sweep and every rounding with it -- three trips there)
    cap = 8 if q < 3 else 3
    elbo, mu, var, it = g.ELBOcalc(max_iter=cap)
    assert g.last_info == 0
    y = np.array(ys)
    Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, g.nodes, g.weights, g.means, g.jitters, y)
    mu0, var0 = cpu_ref.init_mu_var(y, [k.pars[0] for k in g.nodes], [k.pars[0] for k in g.weights], g.jitters)
    e_ref, mu_ref, var_ref, it_ref, _ = cpu_ref.elbo_calc(Kf, Kw, Lf, Lw, yres, y, np.array(es)**2, j2, mu0, var0,
                                                          max_iter=cap, form='ref')
    what = 'random problem %d (N=%d p=%d q=%d)' % (seed, t.size, p, q)
    assert np.isfinite(e_ref), what + ': the oracle itself left the finite numbers'
    assert it == it_ref, what
    # (round 5 had to widen this bound to 4e-16 cond(K) for seed 7 -- a pure Periodic kernel under a diverging state, where
    # m^T K^-1 m dominates the ELBO -- because the set-up's panel steps multiplied by explicit inverses of diagonal blocks;
    # since round 6 the factorisation of a prior matrix substitutes as LAPACK does: csrc/diag_tile.h ACC,
    # profiles/r06_prior_term_accuracy.txt)
    np.testing.assert_allclose(elbo, e_ref, rtol=RTOL, err_msg=what)
    _cases.assert_state(what, mu, mu_ref, var, var_ref)
    # ... and the same problem as a list of two vectors side by side (one tile or more), from the state that call left (its
    # warm start if it converged, each vector's own initial state otherwise), against the one-by-one form from the same state
    if any(m_ is None for m_ in means):
        return                                                 # (get_parameters needs every mean: meanfield.py:199-200)
    x0 = np.array(g.get_parameters(), dtype=float)
    x1 = x0 * (1.0 + 0.01 * np.random.RandomState(seed).standard_normal(x0.size))
    start = (None, None) if g._mu is None else (g._mu.copy(), g._var.copy())
    got = g.nELBO_batch([x0, x1], max_iter=cap)
    want = []
    for x in (x0, x1):
        g2 = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g2.set_components(*_random_problem(seed)[3:7])
        g2._mu, g2._var = start
        want.append(g2.nELBO(x, max_iter=cap))
    np.testing.assert_allclose(got, want, rtol=1e-9, err_msg=what)
    _assert_default_schedule(g._backend())


@pytest.mark.parametrize('n,p,q,kind', [(45, 1, 1, 'SE'), (128, 2, 2, 'QP'), (130, 3, 3, 'QP'), (200, 1, 1, 'SE'), (256, 2, 3, 'QP')])
def test_small_path_matches_launch_schedule(n, p, q, kind):
    """Problems of one tile (N <= 128; of two tiles with option "small_path" = 2) run a half-sweep as ONE launch, one
    workgroup per latent GP (csrc/smalln.hip); the launch schedule of the large problems (option 0) computes the same thing through two dozen
    launches.  Same set-up values, same sweeps, same state: both paths use the same diagonal-block kernel and the same
    order in every reduction, so they differ by the rounding of a few reassociated products at most."""
    t, ys, es = synth.rv_series(n, p)
    spec = synth.component_spec(p, q, kind)
    out = {}
    for small in (1, 0):
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(nodes, weights, means, jit)
        g._backend().option('small_path', 2 * small)         # (2: two tiles too, off by default -- slower there)
        ctx = g._setup_device(nodes, weights, means, jit)
        assert g.last_info == 0
        mu0, var0 = g._initMuVar(nodes, weights, jit)
        ctx.set_muvar(mu0, var0)
        e, parts, info = ctx.sweep(3, commit=True)
        assert info == 0
        out[small] = (e, parts, ctx.get_logdet_K()) + ctx.get_muvar()
        sc = ctx.get_scalars()
        out[small] += (sc['logdetB'], sc['trBinv'], sc['muKmu'], sc['q1'])
        _assert_default_schedule(ctx)
    # (two tiles: L_10 comes from the tile contraction here and from the chain's 16 x 16 kernel there -- other summation
    # orders, and cond(K) ~ 1e8 turns 1e-16 into 1e-9 on chol(K)^-1)
    for name, a, b in zip(('elbo', 'parts', 'logdet K', 'mu', 'var', 'logdet B', 'tr B^-1', 'mu K^-1 mu', 'q1'), out[1], out[0]):
        if name in ('mu', 'var'):
            continue
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11 * max(1.0, float(np.abs(b).max())), err_msg=name)
    _cases.assert_state('small path vs launch schedule N=%d p=%d q=%d' % (n, p, q), out[1][3], out[0][3], out[1][4], out[0][4], tol=1e-9)


@pytest.mark.parametrize('max_iter', [0, 1, 3, 4, 9, 20, 10000])
def test_elbocalc_loop_on_the_device_matches_the_host_loop(max_iter, capsys):
    """ELBOcalc's loop (meanfield.py:626-649) runs on the device for one-tile problems -- k_small_tail applies the stop
    rule, sweeps are enqueued eight at a time ahead of its verdict -- and sweep by sweep with the rule on the host
    otherwise.  Same elboArray, same trip count, same final state for every max_iter: none (only the discarded sweep),
    fewer than the rule needs, exactly at a batch boundary, beyond it, and the reference's default."""
    n, p, q = 60, 2, 2
    t, ys, es = synth.rv_series(n, p)
    spec = synth.component_spec(p, q, 'QP')
    res = {}
    for small in (1, 0):
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(nodes, weights, means, jit)
        g._backend().option('small_path', small)
        E, mu, var, it = g.ELBOcalc(max_iter=max_iter)
        said_max = 'Max iterations reached' in capsys.readouterr().out
        res[small] = (E, mu, var, it, g._elbo_history.copy(), said_max, g._mu is not None)
        assert g.last_info == 0
        # warm start from where the loop ended (meanfield.py:598-607)
        E2, _, _, it2 = g.ELBOcalc(max_iter=max_iter, mu='previous', var='previous')
        res[small] += (E2, it2)
    a, b = res[1], res[0]
    assert a[3] == b[3] and a[5] == b[5] and a[6] == b[6] and a[8] == b[8]
    assert a[4].shape == b[4].shape == (a[3] + 1,)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-9)          # (dozens of sweeps: the two paths' roundings drift apart)
    np.testing.assert_allclose([a[0], a[7]], [b[0], b[7]], rtol=1e-9)
    _cases.assert_state('device loop vs host loop, max_iter %d' % max_iter, a[1], b[1], a[2], b[2], tol=1e-9)
    if max_iter == 10000:
        assert 3 < a[3] < 10000 and not a[5]


@pytest.mark.parametrize('n,p,q,kind,B', [(45, 1, 1, 'SE', 5), (60, 2, 2, 'QP', 12), (128, 3, 2, 'QP', 40)])
def test_nelbo_batch_side_by_side(n, p, q, kind, B):
    """inference.nELBO_batch on a one-tile problem: B evaluations of ELBOcalc at B parameter vectors in ONE stream of
    launches (gprn_elbocalc_batch: grid y = evaluation, each with its own matrices, state, loop and stop rule) against
    the same evaluations one by one from the same starting state -- values, and through them trip counts (a different
    trip count moves the value by the stop rule's 1e-3)."""
    t, ys, es = synth.rv_series(n, p)
    spec = synth.component_spec(p, q, kind)

    def fresh():
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        means = [meanfunc.Constant(0.3 * (i + 1)) for i in range(p)]
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(nodes, weights, means, jit)
        return g

    g = fresh()
    x0 = np.array(g.get_parameters(), dtype=float)
    rng = np.random.RandomState(7)
    sets = [x0 * (1.0 + 0.05 * rng.standard_normal(x0.size)) + 0.01 * rng.standard_normal(x0.size) * (x0 == 0)
            for _ in range(B)]
    # ---- cold: every evaluation from its own _initMuVar state
    got = np.array(g.nELBO_batch(sets))
    assert g.last_info == 0
    gs = fresh()
    want = []
    for x in sets:
        gs.set_parameters(x)
        want.append(-gs.ELBOcalc()[0])                       # mu = var = 'init'
    np.testing.assert_allclose(got, want, rtol=1e-9)
    # ---- warm: all from one converged state (nELBO's mu = 'previous'); the batch leaves the object's parameters at the
    # last vector and a converged state behind, as a run of nELBO calls does
    gs.set_parameters(x0)
    _, mu_w, var_w, _ = gs.ELBOcalc()
    g._mu, g._var = mu_w.copy(), var_w.copy()
    got = np.array(g.nELBO_batch(sets))
    want = []
    for x in sets:
        gs.set_parameters(x)
        want.append(-gs.ELBOcalc(mu=mu_w, var=var_w)[0])
    np.testing.assert_allclose(got, want, rtol=1e-9)
    np.testing.assert_allclose(g.get_parameters(), sets[-1])
    assert g._mu is not None and np.all(np.isfinite(g._mu))
    # ---- and the one-by-one form of the same call is still there
    one_by_one = np.array(g.nELBO_batch(sets[:3], batch=False))
    np.testing.assert_allclose(one_by_one, want[:3], rtol=2e-2)       # (each from its predecessor's state)
    _assert_default_schedule(g._backend())


def test_nelbo_batch_falls_back_when_a_one_tile_problem_has_no_batched_form(capsys):
    """with the small path switched off (option small_path = 0) a one-tile problem has no side-by-side form --
    gprn_elbocalc_batch says GPRN_E_UNSUPPORTED, as include/gprn_hip.h promises (round 5 returned 'internal' / E_ARG from
    the mid path and nELBO_batch raised) -- and nELBO_batch evaluates the list one by one."""
    t, ys, es = synth.rv_series(60, 2)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, synth.component_spec(2, 1, 'SE'))
    g = gpyrn.inference(1, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    g._backend().option('small_path', 0)
    x0 = np.array(g.get_parameters(), dtype=float)
    sets = [x0 * (1.0 + 0.02 * k) for k in range(3)]
    capsys.readouterr()
    got = g.nELBO_batch(sets)
    assert 'evaluations side by side' not in capsys.readouterr().out
    g2 = gpyrn.inference(1, t, *[a for pair in zip(ys, es) for a in pair])
    g2.set_components(*synth.build_components(covfunc, meanfunc, synth.component_spec(2, 1, 'SE')))
    g2._backend().option('small_path', 0)
    np.testing.assert_allclose(got, [g2.nELBO(x) for x in sets], rtol=1e-12)


def test_reduce_finalize_relaxed_handover_keeps_the_bits():
    """VERDICT r5 weak #9 / ADVICE r5: k_reduce_finalize hands the per-element terms of tr B^-1 and log det B to the last
    workgroup of a slot with relaxed agent-scope stores, s_waitcnt vmcnt(0) and a relaxed ticket (csrc/vecops.hip says what
    makes that safe on gfx942 / gfx950; any other target compiles the release / acquire form).  Here against that fenced
    form (option fenced_finalize), bit for bit: 100 sweeps = 200 phases at N = 4096 (16 workgroups per slot, dealt over all
    eight XCDs: the last one to finish reads what fifteen others on other L2s wrote), and the 192-matrix weight phases of a
    batch of 32 evaluations at N = 512."""
    N, p, q, kind = synth.CONFIGS[3]
    t, ys, es = synth.rv_series(N, p)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, synth.component_spec(p, q, kind))
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    runs = []
    for fenced in (0, 1):
        ctx.option('fenced_finalize', fenced)
        ctx.set_muvar(mu0, var0)
        elbo, parts, info = ctx.sweep(100, commit=True)
        assert info == 0
        sc = ctx.get_scalars()
        runs.append((elbo, parts, sc['trBinv'], sc['logdetB']))
    ctx.option('fenced_finalize', 0)
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    _assert_default_schedule(ctx)
    # ... and wide: 32 evaluations side by side, 64 + 192 slots of 2 workgroups each per phase
    meta, d, g = _model('mid_N512_p3q2')
    x = np.array(g.get_parameters(), dtype=float)
    ctx, kp, yr, jt, m0, v0 = _batch_inputs(g, [x * (1.0 + 0.01 * k) for k in range(32)])
    outs = []
    for fenced in (0, 1):
        ctx.option('fenced_finalize', fenced)
        outs.append(ctx.elbocalc_batch(kp, yr, jt, m0, v0, 6, want_state=True))
    ctx.option('fenced_finalize', 0)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][4], outs[1][4])
    assert np.array_equal(outs[0][5], outs[1][5])


def _batch_inputs(g, sets):
    """What inference._nELBO_batch_device hands the library, vector by vector (the general layout), with the starting state
    `g` holds (or each vector's own _initMuVar state)."""
    ctx = g._backend()
    from itertools import chain
    y_raw = np.concatenate(g.y)
    kp, yr, jt, m0, v0 = [], [], [], [], []
    for i, x in enumerate(sets):
        g.set_parameters(x)
        nodes, weights, means, jitters = g._get_components()
        specs = [g._kernel_spec(k) for k in chain(nodes, weights)]
        assert all(sp[0] == 'device' for sp in specs)
        if i == 0:
            for gp, sp in enumerate(specs):
                g._send_spec(ctx, gp, sp)
            g._prior_key = None
        kp.append(np.concatenate([sp[2] for sp in specs]))
        yr.append(y_raw - g._mean(means))
        jt.append(np.asarray(jitters, dtype=float))
        mu, var = (g._mu, g._var) if g._mu is not None else g._initMuVar(nodes, weights, jitters)
        m0.append(np.ravel(mu))
        v0.append(np.ravel(var))
    return ctx, np.array(kp), np.array(yr), np.array(jt), np.array(m0), np.array(v0)


@pytest.mark.parametrize('tag,B', [('mid_N300_p3q2', 7), ('mid_N512_p3q2', 32), ('cfg1_N200', 5), ('mid_N1024_p1q1', 6),
                                   ('illc_N300_p2q3', 4), ('kmix_N200_p2q2', 6), ('illc_N1000_p2q3', 3)])
def test_every_slot_of_a_batch_above_one_tile_reproduces_the_reference(tag, B):
    _every_slot_reproduces_the_reference(tag, B)


@pytest.mark.parametrize('tag,B', [('step_p3q2', 9), ('step_p2q3', 5), ('illc_N100_p2q3', 6)])
def test_every_slot_of_a_one_tile_batch_reproduces_the_reference(tag, B):
    """... and the one-tile form (csrc/smalln.hip: every kernel's grid y is the evaluation), the ill-conditioned prior of
    round 6 among the fixtures: its set-up kernel k_small_prior_b factors K with substitution panels (diag_tile.h ACC)."""
    _every_slot_reproduces_the_reference(tag, B)


def _every_slot_reproduces_the_reference(tag, B):
    """gprn_elbocalc_batch above one tile (csrc/midn.hip): B evaluations go through the launch schedule with its batch
    dimension = evaluations x latent GPs.  Every slot of a batch run AT THE FIXTURE'S PARAMETERS must reproduce what the
    reference itself printed for them: the forced sweeps (max_iter = their number: the loop's trip i is forced sweep i - 1,
    quirk Q7) -- last ELBO and final state to 1e-8 -- and, where the fixture holds it, the whole ELBOcalc (trip count,
    converged value and state).  No fallback to the event schedule."""
    meta, d, g = _model(tag)
    x = np.array(g.get_parameters(), dtype=float)
    ctx, kp, yr, jt, m0, v0 = _batch_inputs(g, [x] * B)
    assert np.array_equal(m0[0], np.ravel(d['mu_init']))
    k = int(meta['nsweeps'])                                   # (no fixture's loop stops before its forced sweeps end)
    res = ctx.elbocalc_batch(kp, yr, jt, m0, v0, k, want_state=True)
    assert res is not None, 'the library has no batched form for this problem'
    elbo, iters, conv, info, mu, var = res
    assert not info.any() and (iters == k).all()
    np.testing.assert_allclose(elbo, np.full(B, d['elbo_sweeps'][k - 1]), rtol=RTOL)
    for b in range(B):
        _cases.assert_state('batch slot %d of %d, forced sweeps %s' % (b, B, tag), mu[b], d['mu_final'], var[b], d['var_final'])
    res1 = ctx.elbocalc_batch(kp, yr, jt, m0, v0, 1, want_state=True)       # (the fixture's first-sweep state pins trip 1)
    for b in range(B):
        _cases.assert_state('batch slot %d of %d, first sweep %s' % (b, B, tag), res1[4][b], d['mu_1'], res1[5][b], d['var_1'])
    # max_iter = 0: only the discarded call -- elboArray[0], no trip, the state as given
    res0 = ctx.elbocalc_batch(kp, yr, jt, m0, v0, 0, want_state=True)
    np.testing.assert_allclose(res0[0], np.full(B, d['elbo_sweeps'][0]), rtol=RTOL)
    assert (res0[1] == 0).all() and np.array_equal(res0[4].reshape(B, -1), m0)
    if 'calc_elbo' in d:
        elbo, iters, conv, info, mu, var = ctx.elbocalc_batch(kp, yr, jt, m0, v0, 10000, want_state=True)
        assert not info.any() and conv.all() and (iters == int(d['calc_iter'])).all()
        np.testing.assert_allclose(elbo, np.full(B, float(d['calc_elbo'])), rtol=RTOL)
        for b in range(B):
            _cases.assert_state('batch slot %d of %d, ELBOcalc %s' % (b, B, tag), mu[b], d['calc_mu'], var[b], d['calc_var'])
        # ... and the warm start from there (meanfield.py:598-607, 1102-1104)
        mw = np.tile(np.ravel(d['calc_mu']), (B, 1))
        vw = np.tile(np.ravel(d['calc_var']), (B, 1))
        elbo, iters, conv, info = ctx.elbocalc_batch(kp, yr, jt, mw, vw, 10000)
        assert (iters == int(d['warm_iter'])).all()
        np.testing.assert_allclose(elbo, np.full(B, float(d['warm_elbo'])), rtol=RTOL)
    _assert_default_schedule(ctx)


@pytest.mark.parametrize('n,p,q,kind,B,budget_mb', [(200, 1, 1, 'SE', 6, 0), (300, 3, 2, 'QP', 9, 0), (512, 3, 2, 'QP', 32, 0),
                                                    (260, 2, 3, 'QP', 5, 0), (300, 2, 2, 'QP', 11, 80), (1024, 1, 1, 'QP', 4, 0)])
def test_nelbo_batch_side_by_side_above_one_tile(n, p, q, kind, B, budget_mb, capsys):
    """inference.nELBO_batch above one tile: B PERTURBED parameter vectors side by side against the same evaluations one by
    one from the same starting state, cold (each from its own _initMuVar state) and warm (all from one converged state) --
    values to 1e-9 and, through them, trip counts; evaluations that stop at different trips leave the launches one by one
    (the tables are compacted); q = 2 and 3 bring the Q1 traces; a memory budget smaller than the list splits it into
    chunks.  No fallback."""
    t, ys, es = synth.rv_series(n, p)
    spec = synth.component_spec(p, q, kind)

    def fresh():
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        means = [meanfunc.Constant(0.3 * (i + 1)) for i in range(p)]
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(nodes, weights, means, jit)
        return g

    g = fresh()
    if budget_mb:
        g._backend().option('batch_mem_mb', budget_mb)        # (a chunk then holds fewer evaluations than the list has)
    max_iter = 6 if q >= 3 else None                           # (the reference's Jacobi iteration diverges at q = 3: DESIGN.md 3)
    x0 = np.array(g.get_parameters(), dtype=float)
    rng = np.random.RandomState(11)
    sets = [x0 * (1.0 + 0.05 * rng.standard_normal(x0.size)) + 0.01 * rng.standard_normal(x0.size) * (x0 == 0)
            for _ in range(B)]
    capsys.readouterr()
    got = np.array(g.nELBO_batch(sets, max_iter=max_iter))
    assert 'evaluations side by side' in capsys.readouterr().out, 'the list was evaluated one by one: no batched form?'
    assert g.last_info == 0 and np.all(np.isfinite(got))
    gs = fresh()
    want, trips = [], []
    for x in sets:
        gs.set_parameters(x)
        e, _, _, it = gs.ELBOcalc(max_iter=max_iter)
        want.append(-e)
        trips.append(it)
    np.testing.assert_allclose(got, want, rtol=1e-9)
    # ---- warm
    gs.set_parameters(x0)
    _, mu_w, var_w, _ = gs.ELBOcalc(max_iter=max_iter)
    g._mu, g._var = mu_w.copy(), var_w.copy()
    got = np.array(g.nELBO_batch(sets, max_iter=max_iter))
    want = []
    for x in sets:
        gs.set_parameters(x)
        e, _, _, it = gs.ELBOcalc(max_iter=max_iter, mu=mu_w, var=var_w)
        want.append(-e)
        trips.append(it)
    np.testing.assert_allclose(got, want, rtol=1e-9)
    np.testing.assert_allclose(g.get_parameters(), sets[-1])
    if q < 3 and B >= 9:
        assert len(set(trips)) > 1, 'every evaluation stopped at the same trip: the compaction was not exercised'
    _assert_default_schedule(g._backend())
    _assert_default_schedule(gs._backend())


def test_a_batch_the_device_cannot_hold_in_one_piece_runs_in_smaller_chunks(capsys):
    """The memory budget of a chunk is an estimate (option "batch_mem_mb", or half of what hipMemGetInfo reports free).  When
    the device cannot give that much -- here: a budget of 600 GB on a 288 GB device, 96 evaluations of BASELINE config 3
    (N = 4096, 8 latent GPs: 4.4 GB each) -- the slabs' allocation fails part-way; the call must halve the chunk until it
    fits, not fail: values against the same evaluations one by one, and the chunk the call ended up with."""
    N, p, q, kind = synth.CONFIGS[3]
    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, kind)

    def fresh():
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(nodes, weights, means, jit)
        return g

    g = fresh()
    ctx = g._backend()
    ctx.option('batch_mem_mb', 600_000)
    g.batch_max_N = N                                          # (nELBO_batch's own size limit is below this problem)
    B = 96
    x0 = np.array(g.get_parameters(), dtype=float)
    rng = np.random.RandomState(5)
    sets = [x0 * (1.0 + 0.02 * rng.standard_normal(x0.size)) for _ in range(B)]
    capsys.readouterr()
    got = np.array(g.nELBO_batch(sets, max_iter=2))
    assert 'evaluations side by side' in capsys.readouterr().out
    assert g.last_info == 0 and np.all(np.isfinite(got))
    chunk = ctx.option('batch_chunk')
    assert 1 <= chunk < B, 'the whole list in one chunk: %d evaluations of 4.4 GB on one device?' % chunk
    gs = fresh()
    for b in (0, chunk - 1, chunk, B - 1):                     # (either side of a chunk boundary)
        gs.set_parameters(sets[b])
        e = gs.ELBOcalc(max_iter=2)[0]
        np.testing.assert_allclose(got[b], -e, rtol=1e-9)
    ctx.option('batch_mem_mb', 64)                             # (the slabs go back before the next test allocates)
    g.nELBO_batch(sets[:2], max_iter=0)
    _assert_default_schedule(ctx)


@pytest.mark.parametrize('n', [77, 497])
def test_nelbo_batch_with_composite_kernels_ragged_sizes_and_a_capped_loop(n, capsys):
    """The side-by-side forms (one tile / above) on what the other batch tests leave out: kernel EXPRESSIONS (a Sum and a
    Multiplication tree: the fill's postfix-program path, each evaluation with its own parameters), a Matern weight (a
    built-in without host-computed reciprocals), sizes that end inside a 16-column block and a tile (77; 497, the
    reference's solar table), non-zero Linear / Constant means, and a max_iter that cuts some loops short while others stop
    by the rule -- against the same evaluations one by one from the same state."""
    p, q = 2, 1
    t, ys, es = synth.rv_series(n, p)

    def fresh():
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        node = covfunc.QuasiPeriodic(1.1, 40.0, 25.0, 0.8) + covfunc.SquaredExponential(0.4, 5.0)
        weights = [covfunc.SquaredExponential(1.0, 70.0) * covfunc.Periodic(1.0, 33.0, 1.2), covfunc.Matern32(0.9, 45.0)]
        g.set_components([node], weights, [meanfunc.Linear(0.001, 0.2), meanfunc.Constant(-0.1)], [0.4, 0.6])
        return g

    g = fresh()
    kernels = list(g.nodes) + list(g.weights)
    if any(k._device_program() is None for k in kernels):
        pytest.skip('a composite of this test has no device program')
    # (Sum / Multiplication keep the reference's quirk: set_parameters updates the composite's own vector, which is what the
    # device program reads -- covfunc.py:56-62)
    x0 = np.array(g.get_parameters(), dtype=float)
    rng = np.random.RandomState(21)
    B = 9
    sets = [x0 * (1.0 + 0.04 * rng.standard_normal(x0.size)) for _ in range(B)]
    for max_iter in (5, None):
        g = fresh()
        capsys.readouterr()
        got = np.array(g.nELBO_batch(sets, max_iter=max_iter))
        assert 'evaluations side by side' in capsys.readouterr().out, 'the list was evaluated one by one: no batched form?'
        assert g.last_info == 0 and np.all(np.isfinite(got))
        gs = fresh()
        want = []
        for x in sets:
            gs.set_parameters(x)
            want.append(-gs.ELBOcalc(max_iter=max_iter)[0])
        np.testing.assert_allclose(got, want, rtol=1e-9)
    _assert_default_schedule(g._backend())


@pytest.mark.parametrize('n,p,q', [(60, 2, 2), (45, 1, 1)])
def test_small_path_after_a_set_up_through_the_launch_schedule(n, p, q):
    """a set-up that ran through the launch schedule (option "small_path" = 0, or gprn_keep_sigma at
    that time) followed by sweeps on the small path -- the switch flipped in between -- used to read a null ticket and, for
    q > 1, a null table of K_j^-1 pointers.  Either set-up now serves either kind of sweep: same values as the all-small
    and the all-launch runs, for gprn_sweep and for gprn_elbocalc without a new set-up."""
    t, ys, es = synth.rv_series(n, p)
    spec = synth.component_spec(p, q, 'QP')

    def run(setup_small, sweep_small, through_elbocalc):
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
        g.set_components(nodes, weights, means, jit)
        ctx = g._backend()
        ctx.option('small_path', setup_small)
        ctx = g._setup_device(nodes, weights, means, jit)
        assert g.last_info == 0
        mu0, var0 = g._initMuVar(nodes, weights, jit)
        ctx.option('small_path', sweep_small)
        if through_elbocalc:
            hist, it, conv, info, mu, var = ctx.elbocalc(7, setup=False, mu=mu0, var=var0)
            assert info == 0
            return np.array(hist), mu, var
        ctx.set_muvar(mu0, var0)
        e, parts, info = ctx.sweep(3, commit=True)
        assert info == 0
        mu, var = ctx.get_muvar()
        return e, mu, var

    for through in (False, True):
        ref = run(1, 1, through)
        for setup_small, sweep_small in ((0, 1), (1, 0), (0, 0)):
            got = run(setup_small, sweep_small, through)
            np.testing.assert_allclose(got[0], ref[0], rtol=1e-9)
            _cases.assert_state('set-up %d / sweeps %d' % (setup_small, sweep_small), got[1], ref[1], got[2], ref[2], tol=1e-9)


@pytest.mark.parametrize('n', [60, 300])
def test_a_failed_evaluation_leaves_a_batch_at_once(n):
    """an evaluation whose factorisation fails (here: a NaN jitter, so no pivot of B is positive) has a
    NaN ELBO that never meets the stop rule; the reference's loop would run to max_iter and return NaN.  The batch returns
    the same -- info > 0, NaN, not converged -- without the 10000 sweeps, and the evaluations beside it are untouched."""
    import time
    p, q, B = 2, 1, 6
    t, ys, es = synth.rv_series(n, p)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, synth.component_spec(p, q, 'QP'))
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    x0 = np.array(g.get_parameters(), dtype=float)
    rng = np.random.RandomState(3)
    sets = [x0 * (1.0 + 0.02 * rng.standard_normal(x0.size)) for _ in range(B)]
    good = np.array(g.nELBO_batch(sets))
    assert np.all(np.isfinite(good)) and g.last_info == 0
    bad = [x.copy() for x in sets]
    bad[2][-1] = np.nan                                        # a jitter: NaN variances, NaN B, no positive pivot
    g._mu = g._var = None
    g2 = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g2.set_components(*synth.build_components(covfunc, meanfunc, synth.component_spec(p, q, 'QP')))
    g2.nELBO_batch(sets[:2])                                   # buffers
    g2._mu = g2._var = None
    t0 = time.perf_counter()
    got = np.array(g2.nELBO_batch(bad))
    dt = time.perf_counter() - t0
    assert np.isnan(got[2]) and g2.last_info > 0
    keep = [0, 1, 3, 4, 5]
    np.testing.assert_allclose(got[keep], good[keep], rtol=1e-12)
    assert dt < 2.0, 'the failed evaluation kept the batch sweeping (%.1f s)' % dt


def test_nelbo_batch_takes_full_length_vectors_with_frozen_parameters():
    """nELBO / set_parameters accept vectors of the free OR the full length (meanfield.py:223-259); with
    frozen parameters the side-by-side form used to raise on the full-length ones instead of evaluating them."""
    n, p, q = 60, 2, 1
    t, ys, es = synth.rv_series(n, p)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, synth.component_spec(p, q, 'QP'))
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    g.ELBOcalc()
    g.freeze_parameter(name='*jitter*')
    full0 = np.array(g.get_parameters(include_frozen=True), dtype=float)
    free = ~g.frozen_mask
    assert free.sum() < full0.size
    rng = np.random.RandomState(5)
    fulls = [full0.copy() for _ in range(4)]
    for x in fulls:
        x[free] *= 1.0 + 0.03 * rng.standard_normal(int(free.sum()))
    mu_w, var_w = g._mu.copy(), g._var.copy()
    a = np.array(g.nELBO_batch(fulls))                          # full length
    g._mu, g._var = mu_w.copy(), var_w.copy()
    b = np.array(g.nELBO_batch([x[free] for x in fulls]))       # free length
    g._mu, g._var = mu_w.copy(), var_w.copy()
    c = np.array(g.nELBO_batch([fulls[0], fulls[1][free], fulls[2], fulls[3][free]]))
    np.testing.assert_allclose(a, b, rtol=1e-13)
    np.testing.assert_allclose(c, b, rtol=1e-13)
    want = []
    for x in fulls:
        g._mu, g._var = mu_w.copy(), var_w.copy()
        want.append(g.nELBO(x))
    np.testing.assert_allclose(b, want, rtol=1e-9)


def test_small_path_reports_a_failed_pivot():
    """jnp.linalg.cholesky semantics on the small path too: a matrix that is not positive definite gives info > 0 (the
    order of the failing minor, LAPACK style) and NaN downstream, no exception (meanfield.py:71-89)."""
    n = 60
    t, ys, es = synth.rv_series(n, 1)

    class Indefinite(covfunc.covFunction):
        _param_names = ('theta',)
        _tag = 'bad'

        def __call__(self, r):
            k = self.pars[0] ** 2 * np.exp(-0.5 * r ** 2 / 30.0 ** 2)
            if k.ndim == 2 and k.shape[0] == k.shape[1]:
                k = k.copy()
                k[40, 40] = -1.0
            return k

    g = gpyrn.inference(1, t, ys[0], es[0])
    g.set_components(Indefinite(1.0), covfunc.SquaredExponential(1.0, 60.0), meanfunc.Constant(0.0), 0.5)
    E, mu, var, it = g.ELBOcalc(max_iter=5)
    assert g.last_info == 41 and not np.isfinite(E)


@pytest.mark.parametrize('n', [200, 300])
def test_launch_path_reports_a_failed_pivot_of_the_set_up(n):
    """The same semantics above one tile (the launch schedule's set-up, unsharded: factor_priors_single): a prior matrix
    that is not positive definite gives info > 0 -- the order of the failing minor -- and NaN downstream, no exception; and
    the object recovers with a proper kernel afterwards."""
    t, ys, es = synth.rv_series(n, 1)

    class Indefinite(covfunc.covFunction):
        _param_names = ('theta',)
        _tag = 'bad'

        def __call__(self, r):
            k = self.pars[0] ** 2 * np.exp(-0.5 * r ** 2 / 30.0 ** 2)
            if k.ndim == 2 and k.shape[0] == k.shape[1]:
                k = k.copy()
                k[150, 150] = -1.0
            return k

    g = gpyrn.inference(1, t, ys[0], es[0])
    g.set_components(Indefinite(1.0), covfunc.SquaredExponential(1.0, 60.0), meanfunc.Constant(0.0), 0.5)
    E, mu, var, it = g.ELBOcalc(max_iter=5)
    assert g.last_info == 151 and not np.isfinite(E)
    g.set_components(covfunc.SquaredExponential(1.0, 30.0), covfunc.SquaredExponential(1.0, 60.0), meanfunc.Constant(0.0), 0.5)
    E2 = g.ELBOcalc()[0]
    assert g.last_info == 0 and np.isfinite(E2)


def test_elboaux_shim_returns_sigma():
    meta, d, g = _model('step_p3q2')
    j2 = np.array(meta['jitters'])**2
    Lf = np.array([np.linalg.cholesky(K) for K in d['Kf']])
    Lw = np.array([np.linalg.cholesky(K) for K in d['Kw']])
    E, mu, var, sF, sW = g.ELBOaux(d['Kf'], d['Kw'], Lf, Lw, d['y_resid'], j2,
                                   d['mu_init'], d['var_init'])
    np.testing.assert_allclose(E, d['elbo_sweeps'][0], rtol=RTOL)
    np.testing.assert_allclose(sF, d['sigmaF_1'], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(sW, d['sigmaW_1'], rtol=1e-6, atol=1e-10)


# (cfg1_N200, mid_N300_p3q2, illc_N300_p2q3, kmix_N200_p2q2: gprn_prior_terms above one tile -- the per-tile-row triangular
# product over chol(K)^-1 and the row-wise dot on a padded leading dimension, ADVICE r5 -- the third with an ill-conditioned
# prior and means far outside its range)
@pytest.mark.parametrize('tag', ['step_p3q2', 'step_p2q3', 'step_p1q1', 'cfg1_N200', 'mid_N300_p3q2', 'illc_N100_p2q3',
                                 'illc_N300_p2q3', 'kmix_N200_p2q2'])
def test_the_four_step_methods_of_inference(tag):
    """inference._updateSigMu / _expectedLogLike / _expectedLogPrior / _entropy (meanfield.py:713, 895, 992, 1069): private in
    the reference, but a method diff of the two classes showed exactly these four missing.  With the reference's
    signatures, on its own inputs -- the prior matrices, the state and the covariances the reference's first sweep produced
    (tests/golden: Kf, Kw, sigmaF_1, sigmaW_1, mu_1) -- against the LogL / LogP / Ent / Sigma it recorded, at 1e-8."""
    meta, d, g = _model(tag)
    q, p, N = meta['q'], meta['p'], meta['N']
    stored = 'Kf' in d                                         # (step_p2q3 -- three nodes -- holds the scalars and the state only)
    if stored:
        Kf, Kw = np.array(d['Kf']), np.array(d['Kw'])
    else:
        Kf = np.array([g._KMatrix(k) for k in g.nodes])
        Kw = np.array([g._KMatrix(k) for k in g.weights])
    Lf = np.array([np.linalg.cholesky(K) for K in Kf])
    Lw = np.array([np.linalg.cholesky(K) for K in Kw])
    j2 = np.array(meta['jitters']) ** 2
    y = np.array(d['y_resid'])
    muF, muW = g._u_to_fhatW(np.ravel(d['mu_init']))
    varF, varW = g._u_to_fhatW(np.ravel(d['var_init']))
    sigma_f, mu_f, sigma_w, mu_w = g._updateSigMu(Kf, Kw, Lf, Lw, y, j2, muF, varF, muW, varW)
    assert sigma_f.shape == (q, N, N) and mu_f.shape == (q, N) and sigma_w.shape == (q, p, N, N) and mu_w.shape == (p, q, N)
    if stored:
        scale = max(np.abs(d['sigmaF_1']).max(), np.abs(d['sigmaW_1']).max())
        np.testing.assert_allclose(sigma_f, d['sigmaF_1'], rtol=1e-7, atol=1e-8 * scale)
        np.testing.assert_allclose(sigma_w, d['sigmaW_1'], rtol=1e-7, atol=1e-8 * scale)
    mu1 = np.array(d['mu_1']).reshape(p + 1, q, N)
    var1 = np.array(d['var_1']).reshape(p + 1, q, N)
    _cases.assert_state('_updateSigMu ' + tag, np.concatenate((mu_f[None], mu_w)), mu1,
                        np.concatenate((np.array([np.diag(S) for S in sigma_f])[None],
                                        np.array([[np.diag(sigma_w[j, i]) for j in range(q)] for i in range(p)]))), var1)
    # the three scalars from the REFERENCE's covariances (where the fixture holds them) and means of that sweep
    logl_ref, logp_ref, ent_ref = d['parts_sweeps'][0]
    sF, sW = (np.array(d['sigmaF_1']), np.array(d['sigmaW_1'])) if stored else (sigma_f, sigma_w)
    mf3, mw = mu1[:1], mu1[1:]
    np.testing.assert_allclose(g._entropy(sF, sW), ent_ref, rtol=RTOL)
    np.testing.assert_allclose(g._expectedLogPrior(Kf, Kw, Lf, Lw, sF, mf3, sW, mw), logp_ref, rtol=RTOL)
    np.testing.assert_allclose(g._expectedLogLike(y, j2, sF, mf3, sW, mw), logl_ref, rtol=RTOL)
    # ... and the object still does what it did before (the shims invalidate the cached set-up)
    E = g.ELBOcalc()[0]
    if 'calc_elbo' in d:
        np.testing.assert_allclose(E, float(d['calc_elbo']), rtol=RTOL)


def test_optimize_runs_and_improves():
    meta, d, g = _model('step_p1q1')
    before = g.ELBOcalc()[0]
    res = g.optimize(vars='jitter1', options={'maxiter': 6})
    assert -res.fun >= before - 1e-9


# --------------------------------------------- full size, size-independent checks
def test_cfg3_properties_full_size():
    """BASELINE config 3 (N=4096, p=3, q=2): the golden first sweeps plus properties
    that need no oracle: ELBO is non-decreasing over converged-direction sweeps'
    tail, variances positive, state finite, repeatability bit for bit."""
    N, p, q, kind = synth.CONFIGS[3]
    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, kind)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(nodes, weights, means, jit)
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    ctx.set_muvar(mu0, var0)
    e_a, parts_a, info = ctx.sweep(3, commit=True)
    assert info == 0 and np.all(np.isfinite(e_a))
    mu, var = ctx.get_muvar()
    assert np.all(var > 0) and np.all(np.isfinite(mu))
    ctx.set_muvar(mu0, var0)
    e_b, _, _ = ctx.sweep(3, commit=True)
    assert np.array_equal(e_a, e_b)                 # deterministic reductions
    if _cases.available('cfg3_N4096'):
        dd = _cases.load('cfg3_N4096')[1]
        np.testing.assert_allclose(e_a[:2], dd['elbo_sweeps'], rtol=RTOL)


def test_overlap_modes_are_bit_identical():
    """What runs beside the factorisations instead of before / behind them (option "overlap": B formed inside the first
    panel's update, the row reductions over X panel by panel, the node term beside the weight phase, log det B inside
    k_finalize, a sweep's end beside the next sweep's node phase) changes WHEN kernels run, never a rounding: four forced
    sweeps of an N = 4096, q = 2 problem (two outer-panel schedules per sweep, T = 32) give the same bits for every mask."""
    N, p, q = 4096, 1, 2
    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, 'QuasiPeriodic')
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(nodes, weights, means, jit)
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    out = {}
    try:
        for mask in (0, 1, 2, 4 | 8, 16, 31):
            ctx.option('overlap', mask)
            ctx.set_muvar(mu0, var0)
            e, parts, info = ctx.sweep(4, commit=True)
            assert info == 0 and np.all(np.isfinite(e))
            out[mask] = (e, parts) + ctx.get_muvar()
        for mask, res in out.items():
            for a, b in zip(res, out[0]):
                assert np.array_equal(a, b), mask
        _assert_default_schedule(ctx)
    finally:
        ctx.option('overlap', 31)


def test_cfg5_size_factorisation_against_lapack():
    """BASELINE config 5's matrix size (N=16384: 128 tiles, 2.1 GB per matrix, offsets beyond
    2^31 bytes) on one node + one weight GP.  No reference value exists at this size, so the
    blocked factor+inverse is checked against LAPACK on the SAME device-filled matrix: log det K,
    and chol(K)^-1 u against a triangular solve; then sweeps must be finite, positive and
    repeatable bit for bit."""
    import scipy.linalg as sl
    N, p, q = 16384, 1, 1
    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, 'QP')
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(nodes, weights, means, jit)
    assert g.last_info == 0
    ld = ctx.get_logdet_K()
    rng = np.random.RandomState(1)
    u = rng.standard_normal(N)
    for gp in (0, 1):                                # node (QuasiPeriodic), weight (SquaredExponential)
        K = ctx.get_matrix(_hip.M_K, gp)
        assert np.array_equal(K, K.T)
        L = np.linalg.cholesky(K)
        np.testing.assert_allclose(ld[gp], 2 * np.log(np.diag(L)).sum(), rtol=1e-10)
        del K
        X = ctx.get_matrix(_hip.M_KLINV, gp)
        want = sl.solve_triangular(L, u, lower=True, check_finite=False)
        np.testing.assert_allclose(X @ u, want, rtol=0, atol=1e-7 * np.abs(want).max())
        del X, L
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    ctx.set_muvar(mu0, var0)
    e_a, _, info = ctx.sweep(2, commit=True)
    assert info == 0 and np.all(np.isfinite(e_a))
    mu, var = ctx.get_muvar()
    assert np.all(var > 0) and np.all(np.isfinite(mu))
    ctx.set_muvar(mu0, var0)
    e_b, _, _ = ctx.sweep(2, commit=True)
    assert np.array_equal(e_a, e_b)
    _assert_default_schedule(ctx)


def test_rccl_calls_on_a_one_rank_communicator(monkeypatch):
    """Every collective of the sharded sweep (grouped row broadcasts, scalar all-reduce,
    barrier) executed for real on RCCL with world = 1: results must not change."""
    from gpyrn_amd import sharding
    monkeypatch.setenv('GPRN_FORCE_RCCL', '1')
    meta, d, g_plain = _model('step_p3q2')
    g_plain._backend().option('small_path', 0)         # (a context with a communicator runs the launch schedule at every size)
    ref = g_plain.ELBOcalc()
    meta, d, g = _model('step_p3q2')

    class OneRank(sharding.Comm):
        def unique_id(self, timeout=0):
            return _hip.comm_unique_id()
    g._comm = OneRank(world=1, rank=0, local_rank=0)
    ctx = g._backend()
    assert ctx.barrier_max(3.5) == 3.5
    out = g.ELBOcalc()
    assert out[3] == ref[3] and out[0] == ref[0]
    assert np.array_equal(out[1], ref[1])


def _run_ranks(module, tag, world, tmp_path, extra_env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(20000 + os.getpid() % 20000),
                   GPRN_COMM_TRANSPORT='shm')
        env.update(extra_env or {})
        out = str(tmp_path / f'rank{r}.npz')
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, '-m', module, tag, out, f'{os.getpid()}_{module}_{tag}_{world}'],
            cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q_ in procs:
                q_.kill()
            raise
        logs.append(o.decode(errors='replace'))
    assert all(pr.returncode == 0 for pr in procs), '\n'.join(logs)
    return [np.load(o) for o in outs]


@pytest.mark.parametrize('tag,world', [('step_p3q2', 2), ('step_p2q3', 3), ('mid_N300_p3q2', 2),
                                       ('mid_N512_p3q2', 4), ('step_p1q1', 3),    # (a rank owning nothing)
                                       ('cfg4_N4096_q4', 4),    # BASELINE config 4 as it is: 16 latent GPs over 4 ranks
                                       # BASELINE config 5's partition features at the largest world a one-GPU box allows
                                       # (its process guard admits six processes on the card, this one included): 15 latent
                                       # GPs, ranks 3 and 4 own no node, ranks 0-1 refactor K_j of later nodes (quirk Q1)
                                       ('cfg5shape_N1024', 5)])
def test_sharded_ranks_on_one_gpu(tag, world, tmp_path):
    """The sharded path of the library itself (owners, helper K_j^-1 factorisations, row
    broadcasts, scalar all-reduce) with `world` processes sharing this box's one GPU.  RCCL
    refuses two ranks on one device, so the collectives travel through the library's host
    shared-memory rehearsal transport (GPRN_COMM_TRANSPORT=shm); everything else is the code
    that runs under RCCL.  Every rank must reproduce the reference's golden values."""
    if not _cases.available(tag):
        pytest.skip('fixture not generated')
    meta, d = _cases.load(tag)
    results = _run_ranks('tests._shard_worker', tag, world, tmp_path)
    for r, res in enumerate(results):
        assert int(res['rank']) == r and int(res['world']) == world and int(res['sw_info']) == 0
        np.testing.assert_allclose(res['sw_elbo'], d['elbo_sweeps'], rtol=RTOL)
        np.testing.assert_allclose(res['sw_parts'], d['parts_sweeps'], rtol=RTOL)
        _cases.assert_state('sharded %s rank %d/%d' % (tag, r, world), res['sw_mu'], d['mu_final'], res['sw_var'], d['var_final'])
        np.testing.assert_allclose(res['sw_mu'], d['mu_final'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(res['sw_var'], d['var_final'], rtol=1e-6, atol=1e-12)
        ld = res['logdet_K']
        np.testing.assert_allclose(ld[:meta['q']], 2 * d['logdiag_Lf'], rtol=1e-9)
        np.testing.assert_allclose(ld[meta['q']:], 2 * d['logdiag_Lw'], rtol=1e-9)
        if 'calc_elbo' in d:
            assert int(res['calc_iter']) == int(d['calc_iter'])
            np.testing.assert_allclose(res['calc_history'], d['calc_elbo_array'], rtol=RTOL)
            _cases.assert_state('sharded ELBOcalc %s rank %d/%d' % (tag, r, world), res['calc_mu'], d['calc_mu'])
            np.testing.assert_allclose(res['calc_mu'], d['calc_mu'], rtol=1e-6, atol=1e-8)
    # every rank holds the same bits (the all-reduce sums in rank order on every rank)
    first = results[0]
    for other in results[1:]:
        assert np.array_equal(first['sw_elbo'], other['sw_elbo'])
        assert np.array_equal(first['sw_mu'], other['sw_mu'])


def test_sharded_fallback_is_rank_coherent(tmp_path):
    """ADVICE r2 / VERDICT r2 #2: an in-kernel dependency wait that gives up on ONE rank only (rank 1's producer
    flag is withheld through the test hook, 20 ms budget).  The sweep issues collectives, so the verdict is
    max-reduced over the ranks before anybody decides: both ranks run the call again on HIP events -- one
    fallback each, flags latched off on both -- and both reproduce the reference's golden values."""
    tag = 'mid_N512_p3q2'
    meta, d = _cases.load(tag)
    if os.environ.get('GPRN_FLAGS') == '0':
        pytest.skip('GPRN_FLAGS=0: the event schedule has no in-kernel waits to time out')
    results = _run_ranks('tests._shard_worker', tag, 2, tmp_path, extra_env={'GPRN_TEST_WITHHOLD_RANK': '1'})
    if any(int(res['flags']) == 1 and int(res['fallbacks']) == 0 for res in results):
        pytest.skip('device-side flags are off on this box (nothing to time out)')
    for res in results:
        assert int(res['sw_info']) == 0
        assert int(res['fallbacks']) == 1 and int(res['flags']) == 0
        np.testing.assert_allclose(res['sw_elbo'], d['elbo_sweeps'], rtol=RTOL)
        np.testing.assert_allclose(res['sw_parts'], d['parts_sweeps'], rtol=RTOL)
        _cases.assert_state('sharded fallback ' + tag, res['sw_mu'], d['mu_final'])
        np.testing.assert_allclose(res['sw_mu'], d['mu_final'], rtol=1e-6, atol=1e-8)
    assert np.array_equal(results[0]['sw_elbo'], results[1]['sw_elbo'])


def test_a_rank_that_fails_its_checks_stops_every_rank(tmp_path):
    """a rank-local error before the collectives of a sweep (rank 1 never set the variational state) used
    to leave the other ranks in their row broadcasts with no time-out.  The verdict of the local checks is now the
    first collective of gprn_sweep / gprn_factor_priors / gprn_predict / gprn_elbocalc: no rank starts the call."""
    results = _run_ranks('tests._shard_worker', 'step_p3q2', 2, tmp_path, extra_env={'GPRN_TEST_BAD_RANK': '1'})
    msgs = {int(r['rank']): str(r['message']) for r in results}
    assert 'set_muvar' in msgs[1]
    assert 'another rank did not pass its checks' in msgs[0]


def test_a_rank_that_dies_inside_a_sweep_does_not_hold_the_others(tmp_path):
    """a collective has no time-out -- a rank that dies inside one left the others holding their GPUs
    until the launcher's 1500 s limit.  Every collective entry point now runs under the library's watchdog (csrc/api.hip):
    two ranks sweep in a loop (shm transport, one GPU), rank 1 is killed from outside once both are under way, and rank 0
    must end ITSELF -- non-zero status, a line that names the entry point, the collective and the rank -- within its budget
    (5 s here; 600 s by default), without restarting anything."""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    budget = 5
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(20000 + os.getpid() % 20000), GPRN_COMM_TRANSPORT='shm',
                   GPRN_TEST_LONG_SWEEPS='100000', GPRN_COMM_BUDGET_S=str(budget))
        out = str(tmp_path / f'rank{r}.npz')
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, '-m', 'tests._shard_worker', 'mid_N300_p3q2', out,
                                       f'{os.getpid()}_watchdog'], cwd=root, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    try:
        t0 = time.time()
        while not all(os.path.exists(o + '.started') for o in outs):
            assert time.time() - t0 < 240, 'the ranks never got under way'
            assert all(pr.poll() is None for pr in procs), 'a rank ended before the test began'
            time.sleep(0.05)
        time.sleep(0.5)                                        # both are inside their loops of sweeps now
        procs[1].send_signal(signal.SIGKILL)
        t_kill = time.time()
        log0, _ = procs[0].communicate(timeout=budget + 60)
        waited = time.time() - t_kill
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    log0 = log0.decode(errors='replace')
    assert procs[0].returncode == 86, log0
    assert 'inside a collective section' in log0 and 'rank 0 of 2' in log0 and 'gprn_sweep' in log0, log0
    assert waited < budget + 10, 'rank 0 held on for %.0f s after rank 1 had died' % waited


def test_a_long_elbocalc_outlives_the_watchdogs_budget(tmp_path):
    """the watchdog's timer was refreshed only by the shm barrier and every 64th sweep of ONE gprn_sweep
    call, so on the RCCL transport a gprn_elbocalc that legitimately ran longer than the budget (N = 16384; a NaN ELBO that
    runs to max_iter) would have been ended on every rank with a line blaming a dead rank.  Every stream synchronisation
    the host observes now restarts the count.  Two ranks, a 1 s budget, one ELBOcalc call of several times that (q = 3: the
    iteration diverges and never meets the stop rule): both ranks must come back, with status 0."""
    budget, max_iter = 1, 12000
    results = _run_ranks('tests._shard_worker', 'step_p2q3', 2, tmp_path,
                         extra_env={'GPRN_TEST_LONG_ELBOCALC': str(max_iter), 'GPRN_COMM_BUDGET_S': str(budget)})
    for res in results:
        assert int(res['iters']) == max_iter
        assert float(res['seconds']) > 1.5 * budget, 'the call ended inside the budget (%.1f s): nothing was tested' % float(res['seconds'])


@pytest.mark.parametrize('tag,world,user', [('step_p3q2', 2, False), ('step_p2q3', 3, False), ('mid_N300_p3q2', 2, True)])
def test_sharded_prediction(tag, world, user, tmp_path):
    """inference._Prediction on a sharded object (meanfield.py:1289-1381; _gp.py:107-138 per latent GP): the owners
    predict their latent GPs, the rows travel as one grouped broadcast, every rank combines them -- and, with
    `user`, the weights are user-defined covFunction subclasses whose K, K* and k** are evaluated on the host of the
    owning rank (gprn_predict_upload).  Every rank against the reference's own prediction."""
    ref = np.load(os.path.join(_cases.GOLDEN, 'pred_' + tag + '.npz'))
    env = {'GPRN_TEST_PREDICT': '1'}
    if user:
        env['GPRN_TEST_USER_WEIGHTS'] = '1'
    for res in _run_ranks('tests._shard_worker', tag, world, tmp_path, extra_env=env):
        assert int(res['pred_info']) == 0
        np.testing.assert_allclose(res['pred_nodes'], ref['node_means'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(res['pred_weights'], ref['weight_means'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(res['pred_mean'], ref['mean'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(res['pred_var'], ref['var'], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('tag', ['mid_N512_p3q2', 'cfg5shape_N2048'])
@pytest.mark.parametrize('env', [{'GPRN_FLAGS': '0'}, {'GPRN_FILL_SYM': '0'}],
                         ids=lambda e: ','.join(f'{k}={v}' for k, v in e.items()))
def test_schedule_and_kernel_variants_agree(env, tag, tmp_path):
    """The two switches of the library that are read once per process -- HIP events instead of device-side flags for
    the factorisation's dependencies (what a serialising tool or a time-out falls back to), and the full-matrix
    covariance fill instead of the symmetric one -- each in a process of its own against the golden values: the latency
    set of task lists (mid_N512_p3q2: T = 4) and the throughput set (BASELINE config 5's shape at N = 2048: T = 16, three
    nodes and twelve weights, four outer panels per factorisation)."""
    if not _cases.available(tag):
        pytest.skip('fixture not generated')
    meta, d = _cases.load(tag)
    res = _run_ranks('tests._shard_worker', tag, 1, tmp_path, extra_env=env)[0]
    assert int(res['sw_info']) == 0 and int(res['fallbacks']) == 0
    # (the worker inherits this process' environment: the whole suite may itself be running under GPRN_FLAGS=0)
    assert int(res['flags']) == (0 if env.get('GPRN_FLAGS', os.environ.get('GPRN_FLAGS')) == '0' else 1)
    np.testing.assert_allclose(res['sw_elbo'], d['elbo_sweeps'], rtol=RTOL)
    np.testing.assert_allclose(res['sw_parts'], d['parts_sweeps'], rtol=RTOL)
    _cases.assert_state('variant %s %s' % (tag, sorted(env.items())), res['sw_mu'], d['mu_final'])
    np.testing.assert_allclose(res['sw_mu'], d['mu_final'], rtol=1e-6, atol=1e-8)


def test_eval_pool_splits_independent_evaluations(tmp_path):
    """sharding.EvalPool: three ranks, each with the whole problem on the (shared)
    GPU, split five nELBO evaluations; every rank gets all five values, bit-identical to
    evaluating them one by one.  nELBO_batch(pool=...) and mcmc(batch=True, pool=...): each rank's share side by
    side, the warm-start state handed round so that every rank holds the same one -- values, state and chain as on
    one GPU without a pool."""
    for res in _run_ranks('tests._pool_worker', 'step_p3q2', 3, tmp_path):
        assert int(res['world']) == 3
        assert np.array_equal(res['pooled'], res['serial'])
        np.testing.assert_allclose(res['one_by_one'], res['serial'], rtol=1e-2)   # (chained: warm starts differ)
        # each rank's share side by side, the state handed round: independent of the number of ranks
        np.testing.assert_allclose(res['batch'], res['serial'], rtol=1e-9)
        np.testing.assert_allclose(res['batch'], res['alone'], rtol=1e-12)
        np.testing.assert_allclose(res['batch_warm'], res['alone_warm'], rtol=1e-12)
        np.testing.assert_allclose(res['mu_pool'], res['mu_alone'], rtol=1e-12, atol=1e-300)
        # mcmc(batch=True, pool=...): the same chain as on one GPU
        np.testing.assert_allclose(res['chain_pool'], res['chain_alone'], rtol=1e-12)
        np.testing.assert_allclose(res['lp_pool'], res['lp_alone'], rtol=1e-9)
        assert np.all(np.isfinite(res['lp_pool'])) and res['chain_pool'].shape[0] == 2
        odd = res['odd']
        assert odd.shape == (7, 2) and np.array_equal(odd[:, 0], np.arange(7.0))
        assert np.isneginf(odd[1, 1]) and np.isnan(odd[2, 1]) and odd[6, 1] == 3.0


# ----------------------------------------------------------------- prediction
@pytest.mark.parametrize('tag', ['step_p1q1', 'step_p3q2', 'step_p2q3', 'cfg1_N200', 'mid_N300_p3q2'])
def test_prediction_matches_reference(tag):
    """inference._Prediction / predict (meanfield.py:1289-1400) on the GPU against the
    reference's own output for the same variational state."""
    meta, d, g = _model(tag)
    ref = np.load(os.path.join(_cases.GOLDEN, 'pred_' + tag + '.npz'))
    mean, var, parts = g._Prediction(tstar=ref['tstar'], mu=d['mu_final'], var=d['var_final'],
                                     separate=True)
    assert g.last_info == 0
    np.testing.assert_allclose(np.array(parts[0], dtype=float), ref['node_means'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(parts[1], dtype=float), ref['weight_means'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mean, ref['mean'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, ref['var'], rtol=1e-6, atol=1e-9)
    # the ELBO path still works afterwards (priors are refactored on demand)
    if 'calc_elbo' in d:
        np.testing.assert_allclose(g.ELBOcalc()[0], float(d['calc_elbo']), rtol=RTOL)


class _UserKernel(covfunc.covFunction):
    """A user-defined covFunction subclass around a built-in: no device program, so K, K* and k** are
    evaluated in Python and handed to the library (gprn_upload_K, gprn_predict_upload)."""

    def __init__(self, inner):
        super().__init__(*inner.pars)
        self._inner = inner
        self._param_names = inner._param_names

    def __call__(self, r):
        return self._inner(r)


@pytest.mark.parametrize('tag', ['step_p1q1', 'step_p3q2', 'step_p2q3', 'cfg1_N200', 'mid_N300_p3q2'])
def test_prediction_with_user_defined_kernels(tag):
    """VERDICT r2 missing #2: the reference's prediction works for ANY covFunction (_predictKMatrix,
    meanfield.py:455-471; _gp.GP.prediction, _gp.py:107-138).  The same fixtures with every node and weight kernel
    swapped for a user subclass: the matrices come from Python, the Cholesky and the solves stay on the GPU."""
    meta, d, g = _model(tag)
    ref = np.load(os.path.join(_cases.GOLDEN, 'pred_' + tag + '.npz'))
    g.set_components([_UserKernel(k) for k in g.nodes], [_UserKernel(k) for k in g.weights], g.means, g.jitters)
    assert g.nodes[0]._device_program() is None
    mean, var, parts = g._Prediction(tstar=ref['tstar'], mu=d['mu_final'], var=d['var_final'], separate=True)
    assert g.last_info == 0
    np.testing.assert_allclose(np.array(parts[0], dtype=float), ref['node_means'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(parts[1], dtype=float), ref['weight_means'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mean, ref['mean'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, ref['var'], rtol=1e-6, atol=1e-9)
    # mixed: built-in nodes, user weights; and a single prediction time (the reference's time.size == 1 branch)
    meta, d, g = _model(tag)
    g.set_components(g.nodes, [_UserKernel(k) for k in g.weights], g.means, g.jitters)
    # (the latent GPs' own predictions and the variance: a mean function such as Linear centres on the mean of the
    # times it is given, meanfunc.py, so the GPRN mean at ONE time is not row 3 of the fixture)
    _, var1, parts1 = g._Prediction(tstar=ref['tstar'][3:4], mu=d['mu_final'], var=d['var_final'], separate=True)
    np.testing.assert_allclose(np.array(parts1[0], dtype=float), ref['node_means'][:, 3:4], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(parts1[1], dtype=float), ref['weight_means'][:, 3:4], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var1, ref['var'][3:4], rtol=1e-6, atol=1e-9)


def test_predict_default_grid_shapes():
    meta, d, g = _model('step_p3q2')
    g.ELBOcalc()
    t, mean, std, parts = g.predict(nn=300)
    assert t.shape == (300,) and mean.shape == (300, 3) and std.shape == (300, 3)
    assert np.all(np.isfinite(mean)) and np.all(std > 0)
    assert np.array(parts[0], dtype=float).shape == (2, 300)


# ------------------------------------------------ reference values at the BASELINE sizes
@pytest.mark.parametrize('tag', ['traj_cfg2_N2048', 'traj_cfg3_N4096'])
def test_elbocalc_trajectory_at_baseline_configs(tag):
    """Full ELBOcalc at BASELINE configs 2 and 3 against the reference's own run (meanfield.py:561-649):
    every ELBOaux value the loop sees, the trip count of the stop rule (:640-646), the converged
    state, and the warm start nELBO uses (:1102-1104)."""
    if not _cases.available(tag):
        pytest.skip('fixture not generated')
    meta, d = _cases.load(tag)
    t, ys, es = synth.rv_series(meta['N'], meta['p'], meta['seed'])
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    g = gpyrn.inference(meta['q'], t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    E, mu, var, it = g.ELBOcalc()
    assert it == int(d['calc_iter'])
    np.testing.assert_allclose(g._elbo_history, d['calc_elbo_array'], rtol=RTOL)
    np.testing.assert_allclose(E, float(d['calc_elbo']), rtol=RTOL)
    _cases.assert_state('ELBOcalc ' + tag, mu, d['calc_mu'], var, d['calc_var'])
    np.testing.assert_allclose(mu, d['calc_mu'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var, d['calc_var'], rtol=1e-6, atol=1e-12)
    E2, mu2, var2, it2 = g.ELBOcalc(mu='previous', var='previous')
    assert it2 == int(d['warm_iter'])
    np.testing.assert_allclose(g._elbo_history, d['warm_elbo_array'], rtol=RTOL)
    _cases.assert_state('warm start ' + tag, mu2, d['warm_mu'])
    np.testing.assert_allclose(mu2, d['warm_mu'], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('tag', ['big_N8192', 'big_N16384'])
def test_reference_sweep_beyond_n4096(tag):
    """One reference-form sweep at N = 8192 / 16384 (p = q = 1; 7.2.N^3 flop on the CPU, generated once
    by oracle/gen_golden.py --big): scalars and the O(N) state of the reference against the HIP path."""
    if not _cases.available(tag):
        pytest.skip('fixture not generated')
    meta, d = _cases.load(tag)
    N = meta['N']
    t, ys, es = synth.rv_series(N, 1, meta['seed'])
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    g = gpyrn.inference(1, t, ys[0], es[0])
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(nodes, weights, means, jit)
    assert g.last_info == 0
    ld = ctx.get_logdet_K()
    np.testing.assert_allclose(ld[:1], 2 * d['logdiag_Lf'], rtol=1e-9)
    np.testing.assert_allclose(ld[1:], 2 * d['logdiag_Lw'], rtol=1e-9)
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    ctx.set_muvar(mu0, var0)
    elbo, parts, info = ctx.sweep(1, commit=True)
    assert info == 0
    np.testing.assert_allclose(elbo, d['elbo_sweeps'], rtol=RTOL)
    np.testing.assert_allclose(parts, d['parts_sweeps'], rtol=RTOL)
    mu, var = ctx.get_muvar()                                    # (2, 1, N): node row, weight row
    _cases.assert_state('reference sweep %s node' % tag, mu[0, 0], d['mu_f_1'][0], var[0, 0], d['var_f_1'][0])
    _cases.assert_state('reference sweep %s weight' % tag, mu[1, 0], np.ravel(d['mu_w_1']), var[1, 0], np.ravel(d['var_w_1']))
    _assert_default_schedule(ctx)
    np.testing.assert_allclose(mu[0, 0], d['mu_f_1'][0], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(mu[1, 0], np.ravel(d['mu_w_1']), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var[0, 0], d['var_f_1'][0], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(var[1, 0], np.ravel(d['var_w_1']), rtol=1e-6, atol=1e-12)


def test_cfg5_full_shape_on_one_gpu():
    """BASELINE config 5 as it is (N = 16384, p = 4, q = 3: 15 latent GPs, quirk Q1 with q = 3, ~130 GB)
    on ONE GPU.  No reference number exists for this shape at this size (7.15.N^3 = 4.6e14 flop per sweep on
    a CPU; the shape itself is pinned to the reference at N = 1024 and 2048, cfg5shape_*), so:
    (1) log det K against the N = 16384 reference fixture where the kernel is the same (node 0 / weight 0 of
        the p = q = 1 problem have identical hyper-parameters and time stamps);
    (2) the B-form identities of the first sweep on node 0 and on weight (0, 0), with NumPy/LAPACK from O(N^2)
        read-backs (K of that latent GP) exactly as oracle/cpu_ref._gp_update_B computes them: diag Sigma,
        Sigma.pred (the new mean), log det B and tr(B^-1) against the per-GP scalars the library reports;
    (3) LogL recomputed from the returned state, Ent and LogP recombined on the host from the per-GP scalars
        (log det K, log det B, tr B^-1, m^T K^-1 m, the cumulative traces of quirk Q1) as cpu_ref.sweep_B does;
    (4) finite, positive variances, bit-repeatable sweeps."""
    from scipy.linalg import solve_triangular
    N, p, q, kind = synth.CONFIGS[5]
    G = q * (p + 1)
    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, kind)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g = gpyrn.inference(q, t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(nodes, weights, means, jit)
    assert g.last_info == 0
    ld = ctx.get_logdet_K()
    assert np.all(np.isfinite(ld))
    if _cases.available('big_N16384'):
        dd = _cases.load('big_N16384')[1]
        # same t (the generator draws t before any output), node 0 = QP(1, 50, 25, 0.7), weight 0 = SE(1, 60)
        np.testing.assert_allclose(ld[0], 2 * dd['logdiag_Lf'][0], rtol=1e-9)
        np.testing.assert_allclose(ld[q], 2 * dd['logdiag_Lw'][0], rtol=1e-9)
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    yraw = np.array(ys)
    yres = yraw - np.array([np.zeros(N) if m is None else m(t) for m in means])
    variance = np.array(jit)[:, None] ** 2 + np.array(es) ** 2

    # ---- one sweep, then the identities on node 0 and weight (0, 0)
    ctx.set_muvar(mu0, var0)
    e1, parts1, info = ctx.sweep(1, commit=True)
    assert info == 0
    mu1, var1 = ctx.get_muvar()
    sc = ctx.get_scalars()
    muF0, muW0 = cpu_ref.split_u(mu0, p, q, N)
    varF0, varW0 = cpu_ref.split_u(var0, p, q, N)

    def b_form(gp, d_vec, pred):
        K = ctx.get_matrix(_hip.M_K, gp)
        s = np.sqrt(d_vec)
        K *= s[:, None]
        K *= s[None, :]
        K[np.diag_indices(N)] += 1.0
        L = np.linalg.cholesky(K)
        del K
        logdetB = 2.0 * np.sum(np.log(np.diag(L)))
        # Sigma pred = D^-1/2 (I - B^-1) D^-1/2 pred, B^-1 v by two triangular solves
        qv = pred / s
        sig_pred = (qv - solve_triangular(L, solve_triangular(L, qv, lower=True), lower=True, trans='T')) / s
        X = solve_triangular(L, np.eye(N), lower=True, overwrite_b=True)
        del L
        binv_diag = np.einsum('ij,ij->j', X, X)
        return (1.0 - binv_diag) / d_vec, sig_pred, logdetB, binv_diag.sum()

    d_n, pred_n = cpu_ref._node_d_and_pred(yres, variance, muF0, muW0, varW0, 0)
    ds, m_new, ldB, trB = b_form(0, d_n, pred_n)
    _cases.assert_state('cfg5 full shape, node 0 vs LAPACK B-form', mu1[0, 0], m_new, var1[0, 0], ds)
    np.testing.assert_allclose(var1[0, 0], ds, rtol=1e-7, atol=1e-14)
    np.testing.assert_allclose(mu1[0, 0], m_new, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sc['logdetB'][0], ldB, rtol=1e-10)
    np.testing.assert_allclose(sc['trBinv'][0], trB, rtol=1e-9)
    d_w, pred_w = cpu_ref._weight_d_and_pred(yres, variance, mu1[0], var1[0], muW0, 0, 0)
    ds, m_new, ldB, trB = b_form(q, d_w, pred_w)
    _cases.assert_state('cfg5 full shape, weight (0,0) vs LAPACK B-form', mu1[1, 0], m_new, var1[1, 0], ds)
    np.testing.assert_allclose(var1[1, 0], ds, rtol=1e-7, atol=1e-14)
    np.testing.assert_allclose(mu1[1, 0], m_new, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sc['logdetB'][q], ldB, rtol=1e-10)
    np.testing.assert_allclose(sc['trBinv'][q], trB, rtol=1e-9)
    # ... and on the node and the weight with the SMALLEST min(d): var = (1 - diag B^-1) / d cancels worst there (VERDICT r4
    # weak #1: the variances' margin to the 1e-8 bound was thinnest where d is small; round 5 took the loss out of the pivots)
    d_nodes = [cpu_ref._node_d_and_pred(yres, variance, muF0, muW0, varW0, j) for j in range(q)]
    j_min = int(np.argmin([dp[0].min() for dp in d_nodes]))
    if j_min != 0:
        ds, m_new, ldB, trB = b_form(j_min, *d_nodes[j_min])
        _cases.assert_state('cfg5 full shape, node %d (smallest min d) vs LAPACK B-form' % j_min, mu1[0, j_min], m_new, var1[0, j_min], ds)
        np.testing.assert_allclose(sc['logdetB'][j_min], ldB, rtol=1e-10)
    del d_nodes
    dmin_w = {}
    for j in range(q):
        for i in range(p):
            dmin_w[(j, i)] = cpu_ref._weight_d_and_pred(yres, variance, mu1[0], var1[0], muW0, j, i)[0].min()
    (jw, iw) = min(dmin_w, key=dmin_w.get)
    if (jw, iw) != (0, 0):
        d_w, pred_w = cpu_ref._weight_d_and_pred(yres, variance, mu1[0], var1[0], muW0, jw, iw)
        ds, m_new, ldB, trB = b_form(q + jw * p + iw, d_w, pred_w)
        _cases.assert_state('cfg5 full shape, weight (%d,%d) (smallest min d = %.1e) vs LAPACK B-form' % (jw, iw, dmin_w[(jw, iw)]),
                            mu1[1 + iw, jw], m_new, var1[1 + iw, jw], ds)
        np.testing.assert_allclose(sc['logdetB'][q + jw * p + iw], ldB, rtol=1e-10)
        np.testing.assert_allclose(sc['trBinv'][q + jw * p + iw], trB, rtol=1e-9)

    # ---- Ent and LogP of that sweep from the per-GP scalars, as oracle/cpu_ref.sweep_B puts them together
    ent = 0.5 * G * N * (1 + cpu_ref.LOG2PI) + 0.5 * np.sum(ld - sc['logdetB'])
    tr = sc['trBinv'].copy()
    for j in range(q):
        tr[j] += sum(sc['q1'][j, k] for k in range(j))             # quirk Q1: + tr(K_j^-1 Sigma_k), k < j
    logp = -0.5 * N * G * cpu_ref.LOG2PI - 0.5 * np.sum(ld) - 0.5 * np.sum(sc['muKmu'] + tr)
    np.testing.assert_allclose(parts1[0, 2], ent, rtol=1e-10)
    np.testing.assert_allclose(parts1[0, 1], logp, rtol=1e-10)
    logl = cpu_ref.expected_loglike(yraw, variance, mu1[0], mu1[1:], var1[0], np.transpose(var1[1:], (1, 0, 2)))
    np.testing.assert_allclose(parts1[0, 0], logl, rtol=RTOL)
    np.testing.assert_allclose(e1, parts1.sum(axis=1) / q, rtol=1e-12)     # meanfield.py:709

    # ---- two sweeps: finite, positive, bit-repeatable
    ctx.set_muvar(mu0, var0)
    e_a, parts_a, info = ctx.sweep(2, commit=True)
    assert info == 0 and np.all(np.isfinite(e_a)) and np.all(np.isfinite(parts_a))
    assert e_a[0] == e1[0]
    mu, var = ctx.get_muvar()
    assert np.all(var > 0) and np.all(np.isfinite(mu))
    np.testing.assert_allclose(e_a, parts_a.sum(axis=1) / q, rtol=1e-12)
    logl = cpu_ref.expected_loglike(yraw, variance, mu[0], mu[1:], var[0], np.transpose(var[1:], (1, 0, 2)))
    np.testing.assert_allclose(parts_a[-1, 0], logl, rtol=RTOL)
    ctx.set_muvar(mu0, var0)
    e_b, parts_b, _ = ctx.sweep(2, commit=True)
    assert np.array_equal(e_a, e_b) and np.array_equal(parts_a, parts_b)
    _assert_default_schedule(ctx)


# ------------------------------------------------ outer-loop callers and the real-data path
def test_optimize_matches_reference_run():
    """inference.optimize (meanfield.py:1114-1152): a 10-iteration Nelder-Mead run of the reference, call
    by call -- same simplex, same nELBO values (warm-started ELBOcalc per evaluation), same result."""
    with open(os.path.join(_cases.GOLDEN, 'opt_N64_p2q1.json')) as f:
        ref = json.load(f)
    t, ys, es = synth.rv_series(ref['N'], ref['p'])
    nodes, weights, means, jit = _cases.components(ref, covfunc, meanfunc)

    def fresh():
        n, w, m, j = _cases.components(ref, covfunc, meanfunc)
        gg = gpyrn.inference(ref['q'], t, *[a for pair in zip(ys, es) for a in pair])
        gg.set_components(n, w, m, j)
        return gg
    g = fresh()
    assert list(g.parameters_dict.keys()) == ref['names']
    np.testing.assert_allclose(g.get_parameters(), ref['x0'], rtol=0, atol=0)
    calls = []
    orig = g.nELBO
    g.nELBO = lambda x, *a, **k: (calls.append([float(orig(x, *a, **k))] + [float(v) for v in x]) or calls[-1][0])
    res = g.optimize(options={'maxiter': 10})
    want = np.array(ref['calls'])
    got = np.array(calls)
    assert got.shape == want.shape and res.nfev == ref['nfev'] and res.nit == ref['nit']
    np.testing.assert_allclose(got[:, 1:], want[:, 1:], rtol=1e-9, atol=1e-12)      # the simplex points
    np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=RTOL)                   # -ELBO at each
    np.testing.assert_allclose(res.x, ref['x'], rtol=1e-9)
    np.testing.assert_allclose(res.fun, ref['fun'], rtol=RTOL)
    # vars='node1.P': everything else frozen
    g2 = fresh()
    r2 = g2.optimize(vars='node1.P', options={'maxiter': 6})
    assert g2.frozen_mask.tolist() == ref['vars_P']['mask']
    np.testing.assert_allclose(r2.x, ref['vars_P']['x'], rtol=1e-9)
    np.testing.assert_allclose(r2.fun, ref['vars_P']['fun'], rtol=RTOL)
    np.testing.assert_allclose(g2.get_parameters(include_frozen=True), ref['vars_P']['all'], rtol=1e-9)


def test_multiconstant_model_matches_reference():
    """ELBOcalc with the per-instrument offsets mean (meanfunc.py:138-187) against the reference's run."""
    d = np.load(os.path.join(_cases.GOLDEN, 'multiconstant.npz'))
    t = d['time']
    g = gpyrn.inference(1, t, d['y'], d['yerr'])
    g.set_components(covfunc.SquaredExponential(1.1, 15.0), covfunc.SquaredExponential(0.9, 40.0),
                     meanfunc.MultiConstant(list(d['offsets']), d['obsid'], t), 0.4)
    assert list(g.parameters_dict.keys()) == [str(s) for s in d['pnames']]
    np.testing.assert_allclose(list(g.parameters_dict.values()), d['pvalues'])
    np.testing.assert_array_equal(g._mean(g.means), d['mean_vec'])
    E, mu, var, it = g.ELBOcalc()
    assert it == int(d['calc_iter'])
    np.testing.assert_allclose(E, float(d['calc_elbo']), rtol=RTOL)
    _cases.assert_state('ELBOcalc multiconstant', mu, d['calc_mu'], var, d['calc_var'])
    np.testing.assert_allclose(mu, d['calc_mu'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var, d['calc_var'], rtol=1e-6, atol=1e-12)


def test_mcmc_matches_reference_chain(monkeypatch, tmp_path):
    """inference.mcmc (meanfield.py:1154-1286) against the reference's own run under the same deterministic
    emcee stand-in (tests/fake_emcee; emcee is not installed here): walkers drawn from the priors, the
    evaluation of the initial walkers (which moves the warm-start state), every stretch move with its
    warm-started, 100-sweep-capped ELBO -- chain, log-probabilities and ELBO blobs number by number."""
    import sys
    import scipy.stats as st
    with open(os.path.join(_cases.GOLDEN, 'mcmc_N48_p1q1.json')) as f:
        ref = json.load(f)
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fake_emcee')
    for name in [m for m in sys.modules if m == 'emcee' or m.startswith('emcee.')]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(fake)
    monkeypatch.chdir(tmp_path)                      # (a real emcee would write gprn.h5 here)
    t, ys, es = synth.rv_series(ref['N'], ref['p'])

    def priors():
        return {'node1.P': st.uniform(15.0, 20.0), 'weight1.ell': st.uniform(30.0, 60.0),
                'jitter1': st.uniform(0.05, 1.5)}
    for label, p0 in (('prior_start', None), ('ellipsoid_start', [23.0, 55.0, 0.45])):
        nodes, weights, means, jit = _cases.components(ref, covfunc, meanfunc)
        g = gpyrn.inference(ref['q'], t, ys[0], es[0])
        g.set_components(nodes, weights, means, jit)
        np.random.seed(ref['seed'])
        sampler = g.mcmc(priors(), p0=p0, vars=list(ref['vars']), niter=ref['niter'])
        want = ref[label]
        assert sampler.iteration == want['iteration'] and g.frozen_mask.tolist() == want['mask']
        np.testing.assert_allclose(sampler.get_chain(), want['chain'], rtol=1e-9)
        np.testing.assert_allclose(sampler.get_blobs(), want['blobs'], rtol=RTOL)       # the ELBO of every walker
        np.testing.assert_allclose(sampler.get_log_prob(), want['log_prob'], rtol=RTOL)
        np.testing.assert_allclose(g.get_parameters(include_frozen=True), want['final_parameters'], rtol=1e-9)


def test_mcmc_with_the_walkers_side_by_side(monkeypatch, tmp_path):
    """inference.mcmc(batch=True): emcee's vectorised log-probability, the walkers of a half-step evaluated side by side
    (nELBO_batch -> gprn_elbocalc_batch), every one from the SAME warm-start state -- where the reference chains them, each
    from its predecessor's converged state (meanfield.py:1102-1104, 1214-1219).  What that changes, pinned:

    An evaluation is at most 100 sweeps under the 1e-3 stop rule, and the rule -- |std / mean| of the last three values,
    :640-643 -- fires when PROGRESS is slow, not when the fixed point is near: its value depends on the state its loop
    started from.  Measured on 40 draws from the priors below (all converged by the rule in every form): side by side
    against chained, median 4.0e-4, 90th percentile 4.1e-3, worst 0.17 relative.  VERDICT r4 expected 2e-3 for every pair
    of converged loops; that does not hold -- for the REFERENCE either: its own chained evaluation of the same 40 vectors in
    the opposite order differs from the first order by median 3.7e-4, 90th percentile 6.4e-3, worst 0.17.  So:
    (1) the same vectors side by side (A), chained in order (B) and chained in reverse order (C), all from one pre-step
        state: every evaluation converged; A against B deviates no more than the reference's chaining deviates from itself
        (B against C: each quantile within a factor 2), its median is inside the rule's tolerance and nine in ten are within 1e-2;
    (2) evaluations that start from the same state agree to 1e-9 whichever way they are run (test_nelbo_batch_side_by_side*);
    (3) the sampler itself runs in both forms from the same seed: finite everywhere, same shapes, and wherever the two
        ensembles still hold the same point their log-probabilities differ like (1) says, not more."""
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fake_emcee'))
    monkeypatch.chdir(tmp_path)
    from scipy import stats
    n, p, q = 48, 1, 1
    t, ys, es = synth.rv_series(n, p)

    def fresh():
        g = gpyrn.inference(q, t, ys[0], es[0])
        g.set_components(covfunc.SquaredExponential(1.0, 20.0), covfunc.SquaredExponential(1.0, 60.0),
                         meanfunc.Constant(0.0), 0.5)
        g.freeze_parameter(name='mean1.c')
        return g

    priors = {'node1.theta': stats.uniform(0.5, 2.0), 'node1.ell': stats.uniform(10.0, 30.0),
              'weight1.theta': stats.uniform(0.5, 2.0), 'weight1.ell': stats.uniform(30.0, 60.0),
              'jitter1': stats.uniform(0.1, 1.0)}
    names = list(priors)
    rng = np.random.RandomState(4)
    X = np.array([[priors[k].rvs(random_state=rng) for k in names] for _ in range(40)])

    g = fresh()
    g.ELBOcalc()
    mu_s, var_s = g._mu.copy(), g._var.copy()

    def side_by_side(points):
        gb = fresh(); gb._mu, gb._var = mu_s.copy(), var_s.copy()
        ctx, kp, yr, jt, m0, v0 = _batch_inputs(gb, [np.array(x) for x in points])
        e, it, conv, info = ctx.elbocalc_batch(kp, yr, jt, m0, v0, 100)
        assert not info.any()
        return np.array(e), np.array(conv, dtype=bool)

    def chained(points):                                       # the reference's way: each from its predecessor's state
        gc = fresh(); gc._mu, gc._var = mu_s.copy(), var_s.copy()
        e, conv = [], []
        for x in points:
            before = gc._mu
            e.append(-gc.nELBO(x, max_iter=100))
            conv.append(gc._mu is not before)                  # (ELBOcalc stores the state on the converged path only, :644-645)
        return np.array(e), np.array(conv, dtype=bool)

    A, ca = side_by_side(X)
    B, cb = chained(X)
    C, cc = chained(X[::-1])
    C, cc = C[::-1], cc[::-1]
    assert ca.all() and cb.all() and cc.all()
    dab, dbc = np.abs(A - B) / np.abs(B), np.abs(B - C) / np.abs(B)
    qs = lambda v: np.array([np.median(v), np.percentile(v, 90), v.max()])
    print('side by side vs chained: median %.2e, 90 %% %.2e, worst %.2e;  chained vs chained in reverse order: %.2e, %.2e, %.2e'
          % (*qs(dab), *qs(dbc)))
    assert np.all(qs(dab) <= 2.0 * qs(dbc) + 1e-6)
    assert np.median(dab) <= 1e-3 and np.percentile(dab, 90) <= 1e-2

    # ---- (3) the sampler in both forms
    chains = {}
    for batch in (True, False):
        np.random.seed(11)
        gm = fresh()
        sampler = gm.mcmc(priors, niter=3, batch=batch)
        chains[batch] = (sampler.get_chain(), sampler.get_log_prob(), sampler.get_blobs())
        assert np.all(np.isfinite(chains[batch][1])) and np.all(np.isfinite(chains[batch][2]))
    assert chains[True][0].shape == chains[False][0].shape == (3, 10, 5)
    same = np.all(chains[True][0] == chains[False][0], axis=2)
    if same.any():
        lb, lc = chains[True][1][same], chains[False][1][same]
        assert np.median(np.abs(lb - lc) / np.abs(lc)) <= 1e-2


def test_kmatrix_and_tiny_nugget_on_device():
    """inference._KMatrix / _tinyNuggetKMatrix (meanfield.py:413-452) called on their own: one fused fill,
    no factorisation, no second context."""
    meta, d, g = _model('step_p3q2')
    t = np.asarray(g.time)
    r = t[:, None] - t[None, :]
    for k in (covfunc.QuasiPeriodic(1.1, 30.0, 12.5, 0.6), covfunc.SquaredExponential(0.9, 7.0) * covfunc.Periodic(1.0, 9.0, 0.8)):
        np.testing.assert_allclose(g._KMatrix(k), k(r) + 1e-6 * np.eye(t.size), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(g._tinyNuggetKMatrix(k), k(r) + 1.25e-12 * np.eye(t.size), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(g._KMatrix(g.nodes[0]), d['Kf'][0], rtol=1e-12, atol=1e-13)
    two = covfunc.Polynomial(1.0, 0.01, 1.5, 2.0)                       # two-argument kernel: no nugget
    np.testing.assert_allclose(g._KMatrix(two), two(t[:, None], t[None, :]), rtol=1e-12)


def test_sample_draws_from_the_prior():
    """inference.sample (meanfield.py:517-539).  The reference draws through scipy's multivariate_normal, so
    the numbers cannot be compared; the construction can: with the generator re-seeded, a draw must be L z for
    the z NumPy hands out and the Cholesky factor L of the tiny-nugget kernel matrix (checked on a kernel
    with a white-noise term, whose factor is well determined), and many draws must carry the prior's variance."""
    meta, d, g = _model('mid_N300_p3q2')
    t = np.asarray(g.time)
    N, q, qp = t.size, g.q, g.q * g.p
    np.random.seed(11)
    fs, ws = g.sample()
    assert fs.shape == (q, N) and ws.shape == (qp, N)
    assert np.all(np.isfinite(fs)) and np.all(np.isfinite(ws))
    k = covfunc.SquaredExponential(1.3, 20.0) + covfunc.WhiteNoise(0.4)
    assert k._device_program() is not None
    np.random.seed(5)
    draw = g._sample_from_gp(k)
    np.random.seed(5)
    z = np.random.standard_normal(N)
    r = t[:, None] - t[None, :]
    L = np.linalg.cholesky(k(r) + 1.25e-12 * np.eye(N))
    np.testing.assert_allclose(draw, L @ z, rtol=1e-9, atol=1e-11)
    # second moment: many draws of one smooth GP have the prior's variance on the diagonal
    np.random.seed(3)
    k = covfunc.SquaredExponential(1.3, 20.0)
    draws = np.array([g._sample_from_gp(k) for _ in range(400)])
    assert abs(draws.var() / 1.3**2 - 1.0) < 0.15


# ------------------------------------------------ analytic gradient (an extension, not parity)
@pytest.mark.parametrize('tag', ['step_p3q2', 'cfg1_N200', 'mid_N300_p3q2'])
def test_grad_elbo_against_finite_differences(tag):
    """inference.grad_ELBO: the N^3 part on the GPU, against central differences of the oracle's ELBO at the
    same FIXED variational state (oracle/cpu_ref.fixed_state_elbo), parameter by parameter: kernel
    hyper-parameters of every node and weight (quirks Q1/Q2 included), mean parameters (zero, quirk Q3),
    jitters."""
    meta, d, g = _model(tag)
    g.ELBOcalc()
    mu_prev, var_prev = g._mu.copy(), g._var.copy()
    E, grad = g.grad_ELBO(mean_sweeps=0)                  # the partial derivative at the fixed state
    assert grad.shape == (len(g.get_parameters(include_frozen=True)),)
    # the same extra sweep through the oracle, with explicit covariances
    t = np.asarray(g.time, dtype=float)
    nodes, weights, means, jit = g.nodes, g.weights, g.means, list(g.jitters)
    Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, nodes, weights, means, jit, g.y)
    E_ref, mu_n, var_n, parts, sig_f, sig_w = cpu_ref.sweep_ref(Kf, Kw, Lf, Lw, yres, g.y, g.yerr2, j2,
                                                               mu_prev, var_prev, return_sigma=True)
    np.testing.assert_allclose(E, E_ref, rtol=RTOL)
    mu_f, mu_w = mu_n[0], mu_n[1:]

    def F():
        Kf_, Kw_, _, _, _, j2_ = cpu_ref.setup(t, nodes, weights, means, jit, g.y)
        return cpu_ref.fixed_state_elbo(Kf_, Kw_, g.y, g.yerr2, j2_, mu_f, mu_w, sig_f, sig_w)

    fd = []
    for k in list(nodes) + list(weights):
        for i in range(k.pars.size):
            v = k.pars[i]
            h = 1e-5 * max(1.0, abs(v))
            k.pars[i] = v + h; up = F()
            k.pars[i] = v - h; dn = F()
            k.pars[i] = v
            fd.append((up - dn) / (2 * h))
    fd += [0.0] * sum(0 if m is None else int(m._parsize) for m in means)
    for i in range(len(jit)):
        v = jit[i]
        h = 1e-5 * max(1.0, abs(v))
        jit[i] = v + h; up = F()
        jit[i] = v - h; dn = F()
        jit[i] = v
        fd.append((up - dn) / (2 * h))
    fd = np.array(fd)
    scale = np.abs(fd).max()
    np.testing.assert_allclose(grad, fd, rtol=2e-5, atol=1e-6 * scale)


def _converged_elbo(g, x, mu0, var0, n_sweeps):
    """ELBO and state after `n_sweeps` forced sweeps from (mu0, var0) at parameter vector x (all parameters)."""
    g.set_parameters(np.array(x, dtype=float))
    nodes, weights, means, jit = g._get_components()
    ctx = g._setup_device(nodes, weights, means, jit)
    ctx.set_muvar(mu0, var0)
    e, _, info = ctx.sweep(n_sweeps, commit=True)
    assert info == 0
    return e[-1], ctx.get_muvar()


def _total_derivative(g, x0, mu0, var0, n_sweeps):
    fd = []
    for i in range(x0.size):
        h = 1e-5 * max(1.0, abs(x0[i]))
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h
        xm[i] -= h
        fd.append((_converged_elbo(g, xp, mu0, var0, n_sweeps)[0] - _converged_elbo(g, xm, mu0, var0, n_sweeps)[0]) / (2 * h))
    return np.array(fd)


def test_grad_elbo_against_differences_of_the_converged_elbo(capsys):
    """VERDICT r2 #7 / missing #4: grad_ELBO against central differences of what `optimize` actually minimises, the
    ELBO the sweeps converge to (here: 60-80 forced sweeps from the `_initMuVar` state, far beyond the stop rule).

    * Zero mean functions (BASELINE config 1): the residual the update reads IS the raw data the likelihood term reads
      (quirk Q3 has nothing to bite on), the converged state is stationary for the reported ELBO, and the partial
      derivative at fixed state is the total derivative: every kernel parameter and the jitter to 1e-5.
    * Non-zero mean functions (step_p2q1: Constant + Linear): the update maximises a bound on y - mean while the ELBO
      is evaluated on y, so the converged state is NOT stationary for it and the envelope theorem does not apply.  The
      mean-function parameters, which have no partial derivative at all, get the documented finite-difference fallback
      and match to 1 %; for the kernel parameters and jitters the test REPORTS the gap between the fixed-state gradient
      and the total derivative (printed; measured 1e-4 ... 0.2 for the large components, O(1) for components an order of
      magnitude smaller) and pins only sign and size of the dominant ones."""
    # ---- zero means
    meta, d, g = _model('cfg1_N200')
    x0 = g.get_parameters(include_frozen=True).copy()
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    _, (mu, var) = _converged_elbo(g, x0, mu0, var0, 60)
    fd = _total_derivative(g, x0, mu0, var0, 60)
    g.set_parameters(x0.copy())
    g._mu, g._var = mu, var
    _, grad = g.grad_ELBO(mean_sweeps=60)
    names = list(g.parameters_dict.keys())
    kernel_like = [i for i, n in enumerate(names) if not n.startswith('mean')]
    np.testing.assert_allclose(grad[kernel_like], fd[kernel_like], rtol=1e-5)
    g._mu, g._var = mu, var
    _, grad_total = g.grad_ELBO(mean_sweeps=60, total=True)    # zero means: nothing is differenced but the mean parameters
    np.testing.assert_allclose(grad_total[kernel_like], fd[kernel_like], rtol=1e-5)
    # ---- non-zero means
    meta, d, g = _model('step_p2q1')
    x0 = g.get_parameters(include_frozen=True).copy()
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    _, (mu, var) = _converged_elbo(g, x0, mu0, var0, 80)
    fd = _total_derivative(g, x0, mu0, var0, 80)
    g.set_parameters(x0.copy())
    g._mu, g._var = mu, var
    _, grad = g.grad_ELBO(mean_sweeps=80)
    _, grad_partial = g.grad_ELBO(mean_sweeps=0)
    names = list(g.parameters_dict.keys())
    is_mean = np.array([n.startswith('mean') for n in names])
    assert np.all(grad_partial[is_mean] == 0.0)
    np.testing.assert_allclose(grad[is_mean], fd[is_mean], rtol=1e-2)
    gap = np.abs(grad - fd) / np.abs(fd)
    with capsys.disabled():
        print('\n   envelope-theorem gap of grad_ELBO at step_p2q1 (fixed-state gradient vs d/dtheta of the converged ELBO):')
        for n, a, b, r in zip(names, grad, fd, gap):
            print(f'      {n:16s} {a: .4e}  {b: .4e}  {r:.1e}')
    big = (~is_mean) & (np.abs(fd) > 0.3 * np.abs(fd[~is_mean]).max())
    assert big.sum() >= 3 and np.all(np.sign(grad[big]) == np.sign(fd[big])) and np.all(gap[big] < 0.3)
    # ---- total=True: where the envelope theorem does not hold every free parameter is differenced, so
    # the gradient handed to optimize(jac=True) has the right sign and size on EVERY component -- the jitter whose
    # fixed-state entry has the wrong sign included
    g._mu, g._var = mu, var
    _, grad_total = g.grad_ELBO(mean_sweeps=80, mean_start=(mu0, var0), total=True)
    assert np.all(np.sign(grad_total) == np.sign(fd))
    np.testing.assert_allclose(grad_total, fd, rtol=5e-2)
    wrong = np.sign(grad) != np.sign(fd)
    assert wrong.any() and names[int(np.argmax(wrong))].startswith('jitter')      # (what total=True is there for)


def test_optimize_with_the_gradient_on_the_recorded_nelder_mead_problem(capsys):
    """optimize(method='L-BFGS-B', jac=True) on the problem of the recorded reference run (opt_N64_p2q1: ten
    Nelder-Mead iterations, two non-zero mean functions).  The objective of the gradient path is the ELBO after a fixed
    number of forced sweeps (smooth in the parameters; the reference's own objective, ELBOcalc under its 1e-3 stop rule,
    is not), its gradient the fixed-state one plus the finite-difference entries of the mean-function parameters --
    exact where the envelope theorem holds, an approximation where quirk Q3 bites (see the test above).  Required: the
    optimiser moves the mean-function parameters (they are variables now), improves the objective it was given, and
    ends at an ELBO -- re-evaluated the reference's way -- no lower than the reference's own run reached."""
    with open(os.path.join(_cases.GOLDEN, 'opt_N64_p2q1.json')) as f:
        ref = json.load(f)
    nodes, weights, means, jit = _cases.components(ref, covfunc, meanfunc)
    t, ys, es = synth.rv_series(ref['N'], ref['p'])
    g = gpyrn.inference(ref['q'], t, *[a for pair in zip(ys, es) for a in pair])
    g.set_components(nodes, weights, means, jit)
    x0 = np.array(ref['x0'])
    np.testing.assert_allclose(g.get_parameters(), x0)
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)
    f0, _ = g.nELBO_and_grad(x0, sweeps=40, start=(mu0, var0))
    g._mu = g._var = None
    res = g.optimize(method='L-BFGS-B', jac=True, sweeps=40, options={'maxiter': 30})
    names = list(g.parameters_dict.keys())
    moved = np.abs(res.x - x0)
    final = g.nELBO(res.x)                               # the reference's objective at the point found
    with capsys.disabled():
        print(f'\n   L-BFGS-B with grad_ELBO: objective {f0:.4f} -> {res.fun:.4f} in {res.nit} iterations ({res.nfev} evaluations); '
              f'nELBO there {final:.4f}; the recorded Nelder-Mead run ended at {ref["fun"]:.4f} after {ref["nfev"]} evaluations')
    assert res.fun < f0 - 1e-3 * abs(f0)
    assert any(moved[i] > 1e-6 for i, n in enumerate(names) if n.startswith('mean'))
    assert final <= ref['fun'] + 1e-3 * abs(ref['fun'])


def test_grad_contraction_on_device_matches_host_contraction():
    """gprn_grad_kernel (K^-1 m, dK/dtheta and the <G, dK> sums all on the GPU: closed forms for SE / Periodic /
    QuasiPeriodic, central differences of the kernel program for the rest, a composite included) against the same
    gradient contracted in NumPy from gprn_grad_matrices' output -- and a user-defined kernel (uploaded matrix)
    falls back to the host, not fails."""
    class MySE(covfunc.covFunction):              # a user kernel: no device program, K is uploaded
        _param_names = ('a', 'l')

        def __call__(self, r):
            return self.pars[0]**2 * np.exp(-0.5 * r**2 / self.pars[1]**2)

    rng = np.random.default_rng(5)
    N, p, q = 300, 2, 3
    t = np.sort(rng.uniform(0, 60, N))
    args = []
    for _ in range(p):
        args += [rng.normal(size=N), rng.uniform(0.1, 0.3, N)]
    g = gpyrn.inference(q, t, *args)
    nodes = [covfunc.SquaredExponential(1.0, 4.0), covfunc.Periodic(1.0, 11.0, 0.8),
             covfunc.QuasiPeriodic(1.0, 20.0, 9.0, 0.7)]
    weights = [covfunc.SquaredExponential(0.8, 15.0), covfunc.Matern32(0.9, 12.0),
               covfunc.RationalQuadratic(0.7, 1.5, 9.0) + covfunc.Cosine(0.3, 7.0), MySE(0.8, 10.0),
               covfunc.Matern52(0.6, 8.0), covfunc.Exponential(0.5, 20.0)]
    g.set_components(nodes, weights, [None] * p, [0.2] * p)
    _, mu0, var0, _ = g.ELBOcalc(max_iter=20)
    nd, wt, mn, jt = g._get_components()
    ctx = g._setup_device(nd, wt, mn, jt)
    ctx.set_muvar(mu0, var0)
    ctx.keep_sigma(True)
    try:
        ctx.sweep(1, commit=True)
        mu, var = ctx.get_muvar()
        seen = []

        def device(gp, m, n):
            out = ctx.grad_kernel(gp, m, n)
            seen.append((gp, out is not None))
            return out

        on_dev = np.array(g._grad_from_state(nd, wt, mn, jt, mu, var, ctx.grad_matrices, device=device))
        on_host = np.array(g._grad_from_state(nd, wt, mn, jt, mu, var, ctx.grad_matrices))
        assert ctx.grad_kernel(q + 3, np.zeros(N), 2) is None        # the user kernel
    finally:
        ctx.keep_sigma(False)
    # every latent GP with a device program went through the device; the user kernel was never offered
    assert [gp for gp, ok in seen if ok] == [0, 1, 2, 3, 4, 5, 7, 8]
    assert on_dev.shape == on_host.shape
    scale = np.abs(on_host).max()
    # closed forms on both sides (the three nodes, the SE weight): to rounding; central differences on both sides
    # (relative step 1e-6 against N^2 terms weighted by K^-1-sized factors): to their common noise
    np.testing.assert_allclose(on_dev[:11], on_host[:11], rtol=2e-6, atol=1e-8 * scale)
    np.testing.assert_allclose(on_dev, on_host, rtol=2e-3, atol=2e-7 * scale)


def test_optimize_with_analytic_gradient():
    """optimize(method='L-BFGS-B', jac=True): runs on the analytic gradient and does not do worse than where
    it started; the gradient of the frozen parameters never reaches scipy."""
    meta, d, g = _model('cfg1_N200')
    before = g.ELBOcalc()[0]
    g.freeze_parameter(name='mean*')
    res = g.optimize(method='L-BFGS-B', jac=True, options={'maxiter': 4})
    assert res.jac.shape == g.get_parameters().shape
    assert -res.fun >= before - 1e-6 * abs(before)
