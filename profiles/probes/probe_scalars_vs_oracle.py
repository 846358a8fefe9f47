"""Per-latent-GP scalars of the first sweep of a fixture -- log det K, log det B, tr B^-1, m^T K^-1 m, the Q1 traces -- from the
device against the CPU oracle's B-form arithmetic (which the reference's golden values pin): which term carries a deviation.
    python profiles/probes/probe_scalars_vs_oracle.py illc_N1000_p2q3"""
import sys
import numpy as np
sys.path.insert(0, '.')
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc
from oracle import cpu_ref
from tests import _cases
tag = sys.argv[1]
meta, d = _cases.load(tag)
p, q, N = meta['p'], meta['q'], meta['N']
G = q * (p + 1)
nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
g = gpyrn.inference(q, np.array(d['time']), *_cases.data_args(d))
g.set_components(nodes, weights, means, jit)
ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
ctx.set_muvar(d['mu_init'], d['var_init'])
e, parts, info = ctx.sweep(1, commit=True)
sc = ctx.get_scalars()
ldK = ctx.get_logdet_K()
# ---- the oracle, term by term
Kf, Kw, Lf, Lw, y, j2 = cpu_ref.setup(d['time'], nodes, weights, means, jit, d['y'])
variance = j2[:, None] + d['yerr'] ** 2
muF, muW = cpu_ref.split_u(d['mu_init'], p, q, N)
varF, varW = cpu_ref.split_u(d['var_init'], p, q, N)
o = {'logdetB': np.zeros(G), 'trBinv': np.zeros(G), 'muKmu': np.zeros(G), 'q1': np.zeros((q, q))}
o_ldK = np.array([2 * np.sum(np.log(np.diag(L))) for L in list(Lf) + list(Lw)])
mu_f, dsf, keep = np.empty((q, N)), np.empty((q, N)), []
for j in range(q):
    dd, pred = cpu_ref._node_d_and_pred(y, variance, muF, muW, varW, j)
    dsf[j], mu_f[j], o['logdetB'][j], o['trBinv'][j], Binv, s = cpu_ref._gp_update_B(Kf[j], dd, pred, need_inverse=(j < q - 1))
    for k, (Bk, sk) in enumerate(keep):
        Sk = (np.eye(N) - Bk) / (sk[:, None] * sk[None, :])
        o['q1'][j, k] = np.sum(cpu_ref.cho_solve((Lf[j], True), np.eye(N)) * Sk)
    keep.append((Binv, s))
state = np.zeros(((p + 1) * q, N))
state[:q] = mu_f
for gp in range(q, G):
    j, i = divmod(gp - q, p)
    dd, pred = cpu_ref._weight_d_and_pred(y, variance, mu_f, dsf, muW, j, i)
    ds, m, o['logdetB'][gp], o['trBinv'][gp], _, _ = cpu_ref._gp_update_B(Kw[gp - q], dd, pred)
    state[(1 + i) * q + j] = m
for gp in range(G):
    L = Lf[gp] if gp < q else Lw[gp - q]
    a = cpu_ref.solve_triangular(L, state[gp], lower=True)
    o['muKmu'][gp] = a @ a
print('ELBO device %.12e fixture %.12e rel %.2e' % (e[0], d['elbo_sweeps'][0], abs(e[0] - d['elbo_sweeps'][0]) / abs(d['elbo_sweeps'][0])))
print('log det K  abs dev', ' '.join('%.1e' % x for x in np.abs(ldK - o_ldK)))
for k in ('logdetB', 'trBinv', 'muKmu'):
    print('%-10s abs dev' % k, ' '.join('%.1e' % x for x in np.abs(sc[k] - o[k])), '| values', ' '.join('%.3e' % x for x in o[k]))
print('q1 abs dev', np.abs(sc['q1'] - o['q1']).ravel(), 'values', o['q1'].ravel())
print('LogP = -0.5 sum(logdetK + muKmu + trBinv + q1): total abs dev of the sum %.3e against |LogP| %.3e' % (
    0.5 * abs(np.sum(ldK - o_ldK) + np.sum(sc['muKmu'] - o['muKmu']) + np.sum(sc['trBinv'] - o['trBinv']) + np.sum(sc['q1'] - o['q1'])), abs(d['parts_sweeps'][0][1])))
mm = ctx.get_muvar()[0].reshape(-1, N)
print('state rows vs oracle', ' '.join('%.1e' % (np.abs(mm[r] - state[r]).max() / np.abs(state[r]).max()) for r in range(G)))
