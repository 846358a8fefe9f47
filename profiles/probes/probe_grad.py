"""grad_ELBO against central differences of the CONVERGED ELBO (many forced sweeps): the envelope-theorem gap."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc
from tests import _cases

for tag, K in (('step_p2q1', 80), ('step_p1q1', 80), ('cfg1_N200', 60)):
    meta, d = _cases.load(tag)
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    g = gpyrn.inference(meta['q'], np.array(d['time']), *_cases.data_args(d))
    g.set_components(nodes, weights, means, jit)
    x0 = g.get_parameters(include_frozen=True).copy()
    mu0, var0 = g._initMuVar(g.nodes, g.weights, g.jitters)

    def F(x, K=K):
        g.set_parameters(x.copy())
        n, w, m, j = g._get_components()
        ctx = g._setup_device(n, w, m, j)
        ctx.set_muvar(mu0, var0)
        e, _, _ = ctx.sweep(K, commit=True)
        return e[-1], ctx.get_muvar()
    e0, (mu, var) = F(x0)
    e1, _ = F(x0, K + 20)
    print(tag, 'ELBO after', K, 'sweeps', e0, 'after', K + 20, e1, 'rel change', abs(e1 - e0) / abs(e0))
    fd = []
    for i in range(x0.size):
        h = 1e-5 * max(1.0, abs(x0[i]))
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h; xm[i] -= h
        fd.append((F(xp)[0] - F(xm)[0]) / (2 * h))
    fd = np.array(fd)
    g.set_parameters(x0.copy())
    g._mu, g._var = mu, var
    E, grad = g.grad_ELBO(mean_sweeps=K)
    names = list(g.parameters_dict.keys())
    for nme, a, b in zip(names, grad, fd):
        print(f'   {nme:14s} grad {a: .6e}  fd(total) {b: .6e}  gap {abs(a - b) / max(abs(b), 1e-12): .2e}')
