"""Where an ensemble step of inference.mcmc(batch=True) spends its time at the solar table's size (N = 497, p = 4, q = 1):
cProfile of ten steps under the tests' emcee stand-in, and the library's own chunk timers (GPRN_BATCH_TIMERS=1, stderr).
usage: python profiles/mcmc_profile.py [steps]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'fake_emcee'))
import numpy as np
import scipy.stats as st

import bench
import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name, N, p, q, kind = [s for s in bench.LATENCY_SHAPES if s[1] == 497][0]
t, ys, es, spec = bench.latency_problem(N, p, q, kind)
g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair])
g.set_components(*synth.build_components(covfunc, meanfunc, spec))
x0 = np.array(g.get_parameters(), dtype=float)
names = [k for k, fz in zip(g.parameters_dict.keys(), g.frozen_mask) if not fz]
pri = {k: st.uniform(min(0.8 * v, 1.2 * v) - 1e-3, abs(0.4 * v) + 2e-3) for k, v in zip(names, x0)}
np.random.seed(5)
sink = io.StringIO()
with contextlib.redirect_stdout(sink):
    g.nELBO(x0)
    g.nELBO_batch([x0 * (1 + 1e-3 * k) for k in range(len(names))])
os.chdir(tempfile.mkdtemp())
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(sink):
    pr.enable()
    sampler = g.mcmc(pri, niter=steps, batch=True)
    pr.disable()
dt = time.perf_counter() - t0
print('%d walkers, %d steps (+2 initial evaluations of the ensemble): %.1f ms per ensemble step' % (2 * len(names), steps, 1e3 * dt / (steps + 2)))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(28)
print(out.getvalue())
