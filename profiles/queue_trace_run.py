#!/usr/bin/env python3
"""Two sweeps of a BASELINE config with the dataflow schedule's trace on (see queue_timeline.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpyrn_amd as gpyrn  # noqa: E402
from gpyrn_amd import covfunc, meanfunc, synth  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, p, q, kind = synth.CONFIGS[cfg]
t, ys, es = synth.rv_series(N, p)
nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, synth.component_spec(p, q, kind))
g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair])
g.set_components(nodes, weights, means, jit)
ctx = g._setup_device(nodes, weights, means, jit)
ctx.set_muvar(*g._initMuVar(nodes, weights, jit))
ctx.profile_read()                       # drops the set-up's records
print(ctx.sweep(2, commit=True)[0])
ctx.profile_read()                       # writes the trace
