"""How much of a deviation in m^T K^-1 m is the kernel MATRIX itself: the device's fill against NumPy's evaluation of the same
kernel (entries differ in their last bits), each factored by LAPACK on the host, same m.
    python profiles/probes/probe_fill_sensitivity.py illc_N1000_p2q3"""
import sys
import numpy as np
from scipy.linalg import solve_triangular
sys.path.insert(0, '.')
import gpyrn_amd as gpyrn
from gpyrn_amd import _hip, covfunc, meanfunc
from oracle import cpu_ref
from tests import _cases
tag = sys.argv[1]
meta, d = _cases.load(tag)
p, q, N = meta['p'], meta['q'], meta['N']
nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
g = gpyrn.inference(q, np.array(d['time']), *_cases.data_args(d))
g.set_components(nodes, weights, means, jit)
ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
Kf, Kw, Lf, Lw, y, j2 = cpu_ref.setup(d['time'], nodes, weights, means, jit, d['y'])
rows = np.asarray(d['mu_1']).reshape(-1, N)
for gp, Kh in enumerate(list(Kf) + list(Kw)):
    Kd = ctx.get_matrix(_hip.M_K, gp)
    rel = np.abs(Kd - Kh) / np.maximum(np.abs(Kh), 1e-300)
    ulp = np.abs(Kd - Kh) / np.spacing(np.abs(Kh))
    m = rows[gp]
    vals = []
    for K in (Kh, Kd):
        a = solve_triangular(np.linalg.cholesky(K), m, lower=True)
        vals.append(float(a @ a))
    print('%d %-18s cond %.1e | K_dev vs K_numpy: max %.1f ulp, mean %.2f ulp, %.0f %% identical | m^T K^-1 m (LAPACK on each): rel diff %.1e' % (
        gp, type((nodes + weights)[gp]).__name__, np.linalg.cond(Kh), ulp.max(), ulp.mean(), 100 * np.mean(Kd == Kh), abs(vals[0] - vals[1]) / abs(vals[0])), flush=True)
