"""The CPU oracle (oracle/cpu_ref.py) against golden vectors produced by the
reference itself (oracle/gen_golden.py).  No GPU needed."""
import numpy as np
import pytest

from oracle import cpu_ref
from gpyrn_amd import covfunc, meanfunc
from tests import _cases

SMALL = ['step_p1q1', 'step_p2q1', 'step_p1q2', 'step_p3q2', 'step_p2q3']
MID = ['cfg1_N200', 'mid_N300_p3q2', 'mid_N512_p3q2', 'mid_N1024_p1q1']


def _problem(tag):
    meta, d = _cases.load(tag)
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    Kf, Kw, Lf, Lw, y, j2 = cpu_ref.setup(d['time'], nodes, weights, means, jit, d['y'])
    return meta, d, nodes, weights, jit, (Kf, Kw, Lf, Lw, y, d['y'], d['yerr']**2, j2)


@pytest.mark.parametrize('tag', SMALL + ['cfg1_N200'])
def test_init_mu_var(tag):
    meta, d, nodes, weights, jit, _ = _problem(tag)
    mu, var = cpu_ref.init_mu_var(d['y'], [n.pars[0] for n in nodes],
                                  [w.pars[0] for w in weights], jit)
    assert np.array_equal(mu, d['mu_init'])
    assert np.array_equal(var, d['var_init'])
    f, w = cpu_ref.split_u(mu, meta['p'], meta['q'], meta['N'])
    assert np.array_equal(f[None], d['mu_init_f'])
    assert np.array_equal(w, d['mu_init_w'])


@pytest.mark.parametrize('form', ['ref', 'B'])
@pytest.mark.parametrize('tag', SMALL + MID)
def test_forced_sweeps(tag, form):
    meta, d, *_, args = _problem(tag)
    if form == 'ref' and meta['N'] > 600:
        pytest.skip('reference-form sweep at this size is bench territory')
    np.testing.assert_allclose(args[4], d['y_resid'], rtol=0, atol=1e-13)
    sweep = cpu_ref.sweep_ref if form == 'ref' else cpu_ref.sweep_B
    mu, var = d['mu_init'], d['var_init']
    for s in range(meta['nsweeps']):
        E, mu, var, parts = sweep(*args, mu, var)
        # the fixtures were produced by LAPACK on K with cond ~1e8; 1e-8 rel is
        # north_star's tolerance for ELBO and posterior means
        np.testing.assert_allclose(E, d['elbo_sweeps'][s], rtol=1e-8)
        np.testing.assert_allclose(parts, d['parts_sweeps'][s], rtol=1e-8)
        if s == 0:
            np.testing.assert_allclose(mu, d['mu_1'], rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(var, d['var_1'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(mu, d['mu_final'], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('form', ['ref', 'B'])
@pytest.mark.parametrize('tag', ['step_p1q1', 'step_p3q2', 'cfg1_N200', 'mid_N300_p3q2'])
def test_elbo_calc_trajectory(tag, form):
    meta, d, *_, args = _problem(tag)
    if 'calc_elbo' not in d:
        pytest.skip('reference produced no finite value for this case')
    E, mu, var, it, hist = cpu_ref.elbo_calc(*args, d['mu_init'], d['var_init'], form=form)
    assert it == int(d['calc_iter'])
    np.testing.assert_allclose(hist, d['calc_elbo_array'], rtol=1e-8)
    np.testing.assert_allclose(E, float(d['calc_elbo']), rtol=1e-8)
    np.testing.assert_allclose(mu, d['calc_mu'], rtol=1e-6, atol=1e-8)
    assert hist[0] == hist[1]          # Q7: first sweep's update is discarded


@pytest.mark.parametrize('tag', ['step_p1q1', 'step_p3q2', 'step_p2q3', 'cfg1_N200', 'mid_N300_p3q2'])
def test_prediction(tag):
    meta, d = _cases.load(tag)
    g = np.load(_cases.GOLDEN + '/pred_' + tag + '.npz')
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    mean, var, nP, wP = cpu_ref.prediction(d['time'], g['tstar'], nodes, weights, means, jit,
                                           d['mu_final'], d['var_final'], meta['p'], meta['q'])
    np.testing.assert_allclose(nP, g['node_means'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(wP, g['weight_means'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mean, g['mean'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, g['var'], rtol=1e-6, atol=1e-9)
