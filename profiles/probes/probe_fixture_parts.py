import sys, numpy as np
sys.path.insert(0, '.')
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc
from tests import _cases
tag = sys.argv[1] if len(sys.argv) > 1 else 'illc_N1000_p2q3'
meta, d = _cases.load(tag)
for acc in (-2, 1, 0):
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    g = gpyrn.inference(meta['q'], np.array(d['time']), *_cases.data_args(d))
    g.set_components(nodes, weights, means, jit)
    g._backend().option('accurate_factor', acc)
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    ctx.set_muvar(d['mu_init'], d['var_init'])
    for s in range(meta['nsweeps']):
        e, parts, info = ctx.sweep(1, commit=True)
        mu, var = ctx.get_muvar()
        pr = d['parts_sweeps'][s]
        line = 'acc %2d sweep %d elbo rel %.2e parts rel %s' % (acc, s, abs(e[0] - d['elbo_sweeps'][s]) / abs(d['elbo_sweeps'][s]), ' '.join('%.2e' % (abs(parts[0][k] - pr[k]) / abs(pr[k])) for k in range(3)))
        if s == 0:
            m1 = np.asarray(d['mu_1']).reshape(-1, meta['N']); v1 = np.asarray(d['var_1']).reshape(-1, meta['N'])
            mm = mu.reshape(-1, meta['N']); vv = var.reshape(-1, meta['N'])
            line += ' | mu rowwise %s | var %s' % (' '.join('%.1e' % (np.abs(mm[i] - m1[i]).max() / np.abs(m1[i]).max()) for i in range(mm.shape[0])),
                                                    ' '.join('%.1e' % (np.abs(vv[i] - v1[i]).max() / np.abs(v1[i]).max()) for i in range(mm.shape[0])))
            sc = ctx.get_scalars()
            line += ' | muKmu ' + ' '.join('%.6e' % x for x in sc['muKmu'])
        print(line, flush=True)
