import sys, numpy as np
sys.path.insert(0, '.')
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, p, q, kind = synth.CONFIGS[cfg]
if len(sys.argv) > 2:
    N, p, q = (int(x) for x in sys.argv[2].split(','))
t, ys, es = synth.rv_series(N, p)
spec = synth.component_spec(p, q, kind)
nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair])
g.set_components(nodes, weights, means, jit)
ctx = g._setup_device(nodes, weights, means, jit)
mu0, var0 = g._initMuVar(nodes, weights, jit)
ctx.set_muvar(mu0, var0)
ctx.sweep(3, commit=True)
sys.stderr.write('==== timed call\n'); sys.stderr.flush()
e, parts, info = ctx.sweep(6, commit=True)
print('elbo', e[-1], info)
