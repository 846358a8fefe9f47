#!/usr/bin/env python3
"""Headline benchmark: ELBO sweeps per second at BASELINE config 3
(N=4096 time stamps, p=3 outputs, q=2 nodes -> 8 latent GPs), fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--no-cpu]

One "step" = one ELBOaux-equivalent sweep (meanfield.py:651-710) with the
priors already factored, inputs resident in HBM.  With --gpus N > 1 the same
problem's latent GPs are sharded over N ranks, one per GPU
(gpyrn_amd/sharding.py) -- strong scaling.  The ranks come either from a
launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (e.g.
`python -m torch.distributed.run`), or, when WORLD_SIZE is not set, from this
script itself: before anything touches the GPU it starts N copies of itself as
child processes, waits for them and relays rank 0's line.  Rank 0 prints one
JSON line.  No torch import anywhere.

The timed region is `--blocks` (default 5) blocks of exactly K steps, each
bracketed by a barrier + device synchronisation and reduced with a MAX over the
ranks; `value` is K / the median block, the spread is reported beside it.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gpyrn_amd import _hip, covfunc, meanfunc, sharding, synth  # noqa: E402
import gpyrn_amd as gpyrn  # noqa: E402

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X datasheet, fp64 matrix (= fp64 vector); see DESIGN.md §6
TILE = 128


def update_flops(T, n_gp, outer=4, split=True):
    """Algorithmic flops of one sweep's bulk-update launches for n_gp latent GPs, tile by tile as
    csrc/factor.hip::ensure_tasks builds them: the K = 512 `k_tile_gemm<64,64,...>` launches on the look-ahead
    stream ("rest" of each outer update: trailing SYRK tiles and inverse rows beyond the next panel; the
    diagonal and sub-diagonal tiles of the panel after next go with the next-panel launch instead).  Diagonal
    SYRK tiles count their lower triangle only.  Returns (bulk, ahead, next): with `split` the tiles the next
    panel's update writes again -- the columns / rows of the panel after next, the diagonal and sub-diagonal tiles
    of the one after that -- are a launch of their own (family 'update_ahead', kernel tag TG_AHEAD); `next` is the
    part the next panel needs first (the "first" / "next" launches on the side streams, tag TG_NEXT)."""
    bulk = ahead = nxt = 0.0
    for k0 in range(0, T, outer):
        k1 = min(T, k0 + outer)
        n1 = min(T, k1 + outer)
        n2 = min(T, n1 + outer)
        n3 = min(T, n2 + outer)
        kw = (k1 - k0) * TILE
        for i in range(k1, T):
            for j in range(k1, i + 1):
                fl = 2.0 * (TILE * (TILE + 1) / 2 if i == j else TILE * TILE) * kw
                if j < n1 and i <= j + 1:
                    continue                                       # kept up to date step by step (K = 128)
                if j < n1 or (j < n2 and i <= j + 1):
                    nxt += fl                                      # "first" / "next" launches (TG_NEXT)
                    continue
                if split and (j < n2 or (j < n3 and i <= j + 1)):
                    ahead += fl
                else:
                    bulk += fl
            fl = k0 * 2.0 * TILE * TILE * kw                       # R_ic, c < k0
            for c in range(k0, k1):
                fl += 2.0 * TILE * TILE * (k1 - c) * TILE          # R_ic, first touch
            if i < n1:
                nxt += fl
                continue
            if split and i < n2:
                ahead += fl
            else:
                bulk += fl
    return bulk * n_gp, ahead * n_gp, nxt * n_gp


def pmc_traffic():
    """HBM bytes per bulk-update launch from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    coalesced reads on gfx950).  Produced by profiles/summarize_pmc.py; None if absent."""
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd + '_pmc_bulk_update.json')) as f:
                return json.load(f)['hbm_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_mfma():
    """MFMA-pipe utilisation of the bulk-update launches from the committed rocprofv3 PMC pass
    (SQ_VALU_MFMA_BUSY_CYCLES against 1024 SIMDs x launch time x the clock GRBM_GUI_ACTIVE gives; kernels
    serialised by counter collection; profiles/summarize_r02.py)."""
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd + '_pmc_mfma_util.json')) as f:
                d = json.load(f)
            return {'mfma_pipe_busy_fraction': d['mfma_pipe_busy_fraction'], 'effective_clock_ghz': d['effective_clock_ghz'],
                    'tflops_serialised': d['tflops_from_mfma_count'], 'fp64_mfma_per_launch': d.get('fp64_mfma_per_launch'),
                    'source': 'profiles/%s_pmc_mfma_util.json' % rnd}
        except (OSError, KeyError, ValueError):
            continue
    return None


def k512_union():
    """Union-time figures of the K = 512 launches from the committed kernel trace (profiles/summarize_r02.py union)."""
    for rnd in ('r06', 'r05', 'r04', 'r03'):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd + '_k512_union.json')) as f:
                d = json.load(f)
            d['source'] = 'profiles/%s_k512_union.json' % rnd
            return d
        except (OSError, ValueError):
            continue
    return None


def sweep_flops(N, p, q):
    """minimal exact sweep, (2G/3 + (q-1)/3) N^3."""
    G = q * (p + 1)
    return (2 * G / 3 + (q - 1) / 3) * float(N)**3


def _blas_build():
    try:
        from threadpoolctl import threadpool_info
        libs = [i for i in threadpool_info() if i.get('user_api') == 'blas']
        return '; '.join('%s %s (%s threads)' % (i.get('internal_api'), i.get('version'), i.get('num_threads'))
                         for i in libs) or 'unknown'
    except Exception:                      # noqa: BLE001 -- informational only
        return 'unknown'


def cpu_baseline(N, p, q, kind):
    """The reference's formulation (oracle/cpu_ref.sweep_ref, 7 N^3 per latent GP) on this box's
    host cores, on the configuration itself: three consecutive sweeps of all G latent GPs (one set-up) with
    the BLAS threads the library picks (all cores), and -- bounded, because 7 G N^3 on one core takes
    minutes -- one sweep of the p = q = 1 problem of the same N (2 of the G GPs) on ONE BLAS thread,
    scaled by G / 2 (every GP costs the same in that formulation)."""
    from oracle import cpu_ref
    G = q * (p + 1)

    def sweeps(pp, qq, n):
        t, ys, es = synth.rv_series(N, pp)
        spec = synth.component_spec(pp, qq, kind)
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        y = np.array(ys)
        Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, nodes, weights, means, jit, y)
        mu, var = cpu_ref.init_mu_var(y, [n_.pars[0] for n_ in nodes], [w.pars[0] for w in weights], jit)
        t0 = time.time()
        for _ in range(n):
            _, mu, var, _ = cpu_ref.sweep_ref(Kf, Kw, Lf, Lw, yres, y, np.array(es)**2, j2, mu, var)
        return time.time() - t0

    n_all = int(os.environ.get('GPRN_CPU_SWEEPS', 3))            # at least three
    dt_all = sweeps(p, q, n_all)
    out = {'value': n_all / dt_all, 'unit': 'sweeps/s', 'cores': os.cpu_count(), 'kind': 'port',
           'sample': f'{n_all} consecutive reference-formulation sweeps of the configuration (N={N}, p={p}, q={q}: all {G} '
                     f'latent GPs, {dt_all:.1f} s) with NumPy/SciPy LAPACK on all host cores',
           'blas': _blas_build(),
           'note': 'value: the BLAS picks its own thread count (OpenBLAS builds cap it, e.g. 64 on a 256-core box, where the '
                   'LU / Cholesky of a 4096 matrix scales badly); by_threads / best_threads: the same formulation with the BLAS held '
                   'to 8, 16, 32 threads and its default -- divide by best_threads.value, not by value'}
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            dt_1 = sweeps(1, 1, 1)
        out['one_thread'] = {'value': 1.0 / (dt_1 * G / 2), 'unit': 'sweeps/s', 'cores': 1,
                             'sample': f'one sweep at N={N}, p=1, q=1 (2 of {G} latent GPs, {dt_1:.1f} s) on one '
                                       f'BLAS thread, scaled by {G}/2'}
        # a baseline a reader can divide by -- the p = q = 1 sweep of the same N with the BLAS held to 8, 16
        # and 32 threads and at its default, each scaled by G / 2 (every latent GP costs the same in that formulation)
        by = {}
        for nt in (8, 16, 32, None):
            if nt is not None and nt > (os.cpu_count() or 1):
                continue
            with threadpool_limits(limits=nt):
                dt = sweeps(1, 1, 1)
            by['default' if nt is None else str(nt)] = {'value': 1.0 / (dt * G / 2), 'seconds_p1q1': dt}
        best = max(by, key=lambda k: by[k]['value'])
        out['by_threads'] = by
        out['best_threads'] = {'threads': best, 'value': by[best]['value'], 'unit': 'sweeps/s',
                               'sample': f'one sweep at N={N}, p=1, q=1 ({by[best]["seconds_p1q1"]:.1f} s) scaled by {G}/2'}
    except ImportError:
        out['one_thread'] = None
    return out


# ---- the regime the reference itself documents (/root/reference/docs/examples/one_dataset.ipynb cell 20:
# "ELBO=-138.69 (took 2.79 ms)" per nELBO at N = 45): small N, many evaluations -- what optimize() and mcmc() do
# (meanfield.py:1095-1152, 1222-1260).  One "evaluation" = inference.nELBO(x) with changed hyper-parameters: fused fills +
# chol(K) + inverses, then the warm-started ELBOcalc loop to the reference's stop rule.
LATENCY_SHAPES = [
    # name, N, p, q, components
    ('notebook', 45, 1, 1, 'notebook'),       # one_dataset.ipynb: Periodic(1, 13, 1) node, SE(1, 50) weight, jitter 0.1
    ('config 1', 200, 1, 1, 'SE'),            # BASELINE config 1
    ('solar-sized', 497, 4, 1, 'QP'),         # the size of the reference's one real dataset (datasets/Solar_observations.txt:
                                              # 497 epochs; RV + three activity indicators as outputs, one node), synthetic values
    ('mid', 512, 3, 2, 'QP'),                 # the shape of tests/golden/mid_N512_p3q2
    ('config 2', 2048, 1, 1, 'QP'),           # BASELINE config 2
]


def latency_problem(N, p, q, kind):
    if kind == 'notebook':
        # the notebook's own data recipe (cells 6-10) with NumPy's legacy generator; the noise is drawn with
        # RandomState.normal instead of scipy.stats.norm(...).rvs(): same distribution, the timing does not care
        rng = np.random.RandomState(43)
        t = np.sort(rng.uniform(10, 60, N))
        y = 1.5 * np.sin(2 * np.pi * t / 13.5) * np.polyval([0.01, 0.02, 2.5], t)
        yerr = rng.uniform(2, 5, size=N)
        y = y + rng.normal(0.0, np.hypot(0.5, yerr))
        spec = ([('Periodic', [1.0, 13.0, 1.0])], [('SquaredExponential', [1.0, 50.0])], [('Constant', [0.0])], [0.1])
        return t, [y], [yerr], spec
    t, ys, es = synth.rv_series(N, p)
    return t, ys, es, synth.component_spec(p, q, kind)


def latency_cpu(t, ys, es, spec, xs, budget_s=20.0):
    """The same sequence of evaluations through the oracle's reference formulation on the host cores
    (oracle/cpu_ref: K fill + chol(K), then the warm-started loop), at most `budget_s` seconds of them."""
    from oracle import cpu_ref
    y = np.array(ys)
    mu = var = None
    n_done, trips, t0 = 0, [], time.perf_counter()
    for x in xs:
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        rest = np.asarray(x, dtype=float)
        for comp in nodes + weights + means:
            rest = comp.set_parameters(rest)
        jit = list(rest)
        Kf, Kw, Lf, Lw, yres, j2 = cpu_ref.setup(t, nodes, weights, means, jit, y)
        if mu is None:
            mu, var = cpu_ref.init_mu_var(y, [n_.pars[0] for n_ in nodes], [w.pars[0] for w in weights], jit)
        _, mu, var, it, _ = cpu_ref.elbo_calc(Kf, Kw, Lf, Lw, yres, y, np.array(es)**2, j2, mu, var, form='ref')
        n_done += 1
        trips.append(it)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {'value': n_done / dt, 'unit': 'evaluations/s', 'ms_per_evaluation': 1e3 * dt / n_done, 'cores': os.cpu_count(),
            'kind': 'port', 'loop_trips_mean': float(np.mean(trips)),
            'sample': '%d evaluations (reference formulation, NumPy/SciPy LAPACK, %s)' % (n_done, _blas_build())}


def latency(a):
    """python bench.py --latency: nELBO evaluations per second at small N, one JSON line per shape."""
    for name, N, p, q, kind in LATENCY_SHAPES:
        if a.latency_only and str(N) not in a.latency_only.split(','):
            continue
        t, ys, es, spec = latency_problem(N, p, q, kind)
        nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
        g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair])
        g.set_components(nodes, weights, means, jit)
        x0 = np.array(g.get_parameters(), dtype=float)
        reps = a.latency_reps if a.latency_reps > 0 else (200 if N <= 512 else 40)
        # a walk around the starting point: every evaluation has new hyper-parameters (refill + refactor), the state is
        # warm-started from the previous one, as scipy's simplex steps and emcee's walkers do it
        rng = np.random.RandomState(1)
        xs = [x0 * (1.0 + 0.01 * rng.standard_normal(x0.size)) for _ in range(reps)]
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):          # nELBO prints its progress line, as the reference does
            g.nELBO(x0)                                          # allocations, code objects, first factors
            g.nELBO(xs[0])
            ctx = g._backend()
            trips = []
            t0 = time.perf_counter()
            for x in xs:
                g.nELBO(x)
                trips.append(len(g._elbo_history) - 1)
            dt = time.perf_counter() - t0
        # the same walk's points as ONE call: side by side on the device (nELBO_batch -> gprn_elbocalc_batch): one tile -- a
        # half-sweep of all evaluations is one launch (smalln.hip); above -- the launch schedule with its batch dimension =
        # evaluations x latent GPs (midn.hip)
        side = None
        if N <= g.batch_max_N and not a.no_side:
            with contextlib.redirect_stdout(io.StringIO()):
                nb = a.latency_batch if a.latency_batch > 0 else (256 if N <= 256 else 32)
                xb = [x0 * (1.0 + 0.01 * rng.standard_normal(x0.size)) for _ in range(nb)]
                g.nELBO_batch(xb)                                # buffers (sized by the list)
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    vals = g.nELBO_batch(xb)
                    dtb = time.perf_counter() - t0
                    best = dtb if best is None else min(best, dtb)
            side = {'value': nb / best, 'unit': 'evaluations/s', 'evaluations': nb, 'ms_total': 1e3 * best,
                    'vs_one_by_one': (nb / best) / (reps / dt),
                    'all_finite': bool(np.all(np.isfinite(vals))),
                    'schedule': {'flags': int(ctx.option('flags')), 'fallbacks': int(ctx.option('fallbacks'))},
                    'note': 'inference.nELBO_batch: every evaluation with its own matrices, state, loop and stop rule, all in the '
                            'same launches; what an optimiser population or emcee walkers ask for (best of 3 calls)'}
        # ... and what a user of the reference sees of it: inference.mcmc (meanfield.py:1154-1286) for a few ensemble steps, the
        # walkers one by one as the reference evaluates them (:1214-1260) and side by side (batch=True), under the small
        # deterministic emcee stand-in of the tests (emcee is installed nowhere here) with flat priors around the start
        walk = None
        if a.latency_mcmc and N <= 512:
            fake = os.path.join(ROOT, 'tests', 'fake_emcee')
            if os.path.isdir(fake):
                import scipy.stats as st
                sys.path.insert(0, fake)
                names = [k for k, fz in zip(g.parameters_dict.keys(), g.frozen_mask) if not fz]
                pri = {k: st.uniform(min(0.8 * v, 1.2 * v) - 1e-3, abs(0.4 * v) + 2e-3) for k, v in zip(names, x0)}
                walk = {'walkers': 2 * len(names), 'steps': a.latency_mcmc}
                for label, flag in (('one_by_one', False), ('side_by_side', True)):
                    gm = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair])
                    gm.set_components(*synth.build_components(covfunc, meanfunc, spec))
                    np.random.seed(5)
                    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                        gm.nELBO(x0)
                        if flag:
                            gm.nELBO_batch([x0 * (1 + 1e-3 * k) for k in range(len(names))])     # buffers
                        cwd = os.getcwd()
                        import tempfile
                        with tempfile.TemporaryDirectory() as tmp:
                            os.chdir(tmp)                        # (a real emcee would write gprn.h5 here)
                            try:
                                t0 = time.perf_counter()
                                sampler = gm.mcmc(pri, niter=a.latency_mcmc, batch=flag)
                                dtm = time.perf_counter() - t0
                            finally:
                                os.chdir(cwd)
                    evals = walk['walkers'] * (2 + a.latency_mcmc)   # initial walkers (twice: mcmc's and the sampler's) + one per step
                    walk[label] = {'s_total': dtm, 'ms_per_ensemble_step': 1e3 * dtm / (2 + a.latency_mcmc), 'evaluations': evals,
                                   'all_finite': bool(np.all(np.isfinite(sampler.get_log_prob())))}
                walk['speedup'] = walk['one_by_one']['s_total'] / walk['side_by_side']['s_total']
                walk['note'] = 'inference.mcmc under tests/fake_emcee (Goodman-Weare stretch move on NumPy\'s generator), flat priors +-20 % around the start'
                sys.path.remove(fake)
        cpu = None
        if not a.no_cpu:
            # the CPU walk starts at x0 too.  Twice: with ONE BLAS thread (at these sizes the threads of a
            # multi-threaded BLAS cost more than they give: SURVEY.md 6 measured 17 vs 120 sweeps/s at N = 200) and with
            # the threads the library picks; the faster one is the baseline, the other is reported beside it
            g_x = [x0] + xs
            legs = {}
            try:
                from threadpoolctl import threadpool_limits
                with threadpool_limits(limits=1):
                    legs['one_thread'] = latency_cpu(t, ys, es, spec, g_x, budget_s=a.latency_cpu_s / 2)
                legs['one_thread']['cores'] = 1
            except ImportError:
                pass
            legs['all_threads'] = latency_cpu(t, ys, es, spec, g_x, budget_s=a.latency_cpu_s / 2)
            best = max(legs, key=lambda k: legs[k]['value'])
            cpu = dict(legs[best], threads=best, other={k: v for k, v in legs.items() if k != best})
        print(json.dumps({
            'metric': 'nELBO evaluations/sec (changed hyper-parameters, warm start, reference stop rule)',
            'value': reps / dt, 'unit': 'evaluations/s', 'ms_per_evaluation': 1e3 * dt / reps,
            'evaluations': reps, 'loop_trips_mean': float(np.mean(trips)),
            'n_gpus': 1, 'dtype': 'f64', 'data': 'synthetic', 'higher_is_better': True,
            'config': {'workload': '%s: N=%d, p=%d, q=%d' % (name, N, p, q), 'latent_gps': q * (p + 1),
                       'parameters': int(x0.size)},
            'schedule': {'flags': int(ctx.option('flags')), 'fallbacks': int(ctx.option('fallbacks'))},
            'reference_note': ('one_dataset.ipynb cell 20 prints 2.79 ms per nELBO for this problem shape on its author\'s '
                               'machine (jax on CPU)' if kind == 'notebook' else None),
            'side_by_side': side,
            'mcmc': walk,
            'cpu_baseline': cpu}), flush=True)


def golden_check(ctx, a, N, p, q, t, ys):
    """The reference's own numbers for THIS workload, where tests/golden holds them (configs 1-4: forced sweeps from the
    initial state, recorded from /root/reference by oracle/gen_golden.py): the first sweeps of the bench's context against
    them, uncommitted, before the timed region.  `elbo_last` further down is the state after hundreds of sweeps and says
    nothing by itself (at q >= 3 the reference's own iteration diverges: DESIGN.md 3); this does."""
    tag = {1: 'cfg1_N200', 2: 'cfg2_N2048', 3: 'cfg3_N4096', 4: 'cfg4_N4096_q4'}.get(a.config)
    if a.config == 5:
        tag = 'cfg5shape_N%d' % N                           # config 5's shape at a reduced N (rehearsals): the reference's own sweeps
    path = os.path.join(ROOT, 'tests', 'golden', '%s.npz' % tag) if tag and not a.shape else None
    if not path or not os.path.exists(path):
        return None
    d = np.load(path)
    with open(path[:-4] + '.json') as f:
        meta = json.load(f)
    spec = synth.component_spec(p, q, synth.CONFIGS[a.config][3])
    flat = lambda items: [[name, [float(x) for x in pars]] for name, pars in items]
    same_model = all(flat(x) == flat(meta[k]) for x, k in zip(spec[:3], ('nodes', 'weights', 'means'))) and \
        [float(j) for j in spec[3]] == [float(j) for j in meta['jitters']]
    if not (same_model and np.array_equal(d['time'], t) and np.array_equal(d['y'], np.array(ys))):
        return {'fixture': 'tests/golden/%s.npz' % tag, 'same_inputs': False}
    ref = np.asarray(d['elbo_sweeps'], dtype=float)
    elbo, parts, info = ctx.sweep(len(ref), commit=False)
    return {'fixture': 'tests/golden/%s.npz' % tag, 'same_inputs': True, 'sweeps': int(len(ref)), 'info': int(info),
            'elbo_rel_err': float(np.max(np.abs(elbo - ref) / np.abs(ref))),
            'parts_rel_err': float(np.max(np.abs(parts - d['parts_sweeps']) / np.abs(d['parts_sweeps']))),
            'bound': 1e-8}


# what DESIGN.md 6 expects of a BASELINE multi-GPU config in its stated topology, from the partition alone (exchange cost not
# included): the per-rank share of latent GPs measured as an ad-hoc shape on one GPU against the whole config on one GPU
EXPECTED_CEILING = {
    4: {'n_gpus': 4, 'sweeps_per_s': 195.0, 'vs_one_gpu': 3.2,
        'why': '4 nodes + 12 weights over 4 ranks = 1 + 3 latent GPs each: that shape alone runs at 195 sweeps/s against 61.5 for all 16 on one GPU'},
    5: {'n_gpus': 8, 'sweeps_per_s': 7.0, 'vs_one_gpu': 5.0,
        'why': '3 nodes leave five GPUs idle in the node phase and 12 weights split 2/2/2/2/1/1/1/1: the weight phase takes the time of two matrices where one GPU (1.40 sweeps/s) takes twelve'},
}


def config_in_topology(cfg, world, rank, n_override=None, steps=5, blocks=3, warmup=1):
    """BASELINE config `cfg` sharded over the ranks of THIS run -- the topology BASELINE.json states for it when world is 4
    (config 4) or 8 (config 5): its own communicator, set-up, the reference's golden first sweeps where a fixture holds them,
    then timed blocks of forced sweeps (barrier + device sync on both sides, MAX over ranks).  Collective: every rank calls
    it; rank 0 gets the block for `configs_at_n_gpus`, the others None."""
    N, p, q, kind = synth.CONFIGS[cfg]
    if n_override:
        N = int(n_override)
    comm = sharding.Comm()
    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, kind)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair], comm=comm)
    g.set_components(nodes, weights, means, jit)
    t0 = time.time()
    ctx = g._setup_device(nodes, weights, means, jit)
    ctx.barrier_max(0.0)
    t_setup = time.time() - t0
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    ctx.set_muvar(mu0, var0)

    class _A:                                              # what golden_check reads of the command line
        config, shape = cfg, (None if (not n_override or cfg == 5) else 'x')
    golden = golden_check(ctx, _A, N, p, q, t, ys)         # (a sweep of a sharded context: every rank takes part)
    if warmup > 0:
        ctx.sweep(warmup, commit=True)
    block_s = []
    for _ in range(max(1, blocks)):
        ctx.barrier_max(0.0)
        t0 = time.perf_counter()
        elbo, parts, info = ctx.sweep(steps, commit=True)
        block_s.append(ctx.barrier_max(time.perf_counter() - t0))
    dt = float(np.median(block_s))
    out = None
    if rank == 0:
        own = sharding.owners(p, q, world)
        out = {'config': cfg, 'workload': 'BASELINE config %d: N=%d, p=%d outputs, q=%d nodes%s' % (
                   cfg, N, p, q, '' if not n_override else ' (N reduced for a rehearsal)'),
               'value': steps / dt, 'unit': 'sweeps/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'blocks': len(block_s),
               'sweeps_per_s': {'median': steps / dt, 'min': steps / max(block_s), 'max': steps / min(block_s)},
               'scaling': 'strong', 'dtype': 'f64', 'data': 'synthetic',
               'latent_gps': q * (p + 1), 'latent_gps_per_rank': [own.count(r) for r in range(world)],
               'nodes_per_rank': [own[:q].count(r) for r in range(world)],
               'comm_ranks': ctx.world, 'transport': os.environ.get('GPRN_COMM_TRANSPORT', 'rccl'),
               'sweep_tflops': sweep_flops(N, p, q) * steps / dt / 1e12,
               'sweep_frac_of_n_gpu_peak': sweep_flops(N, p, q) * steps / dt / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
               'setup_s': t_setup, 'golden_check': golden, 'elbo_last': float(elbo[-1]), 'info': int(info),
               'schedule': {'flags': int(ctx.option('flags')), 'fallbacks': int(ctx.option('fallbacks'))},
               'expected_ceiling': EXPECTED_CEILING.get(cfg) if not n_override else None}
    ctx.barrier_max(0.0)
    comm.cleanup()
    g._ctx.close() if getattr(g, '_ctx', None) is not None else None
    return out


def self_launch(a):
    """--gpus N > 1 without a launcher: N child processes, one per rank, started BEFORE this
    process makes any HIP call (a process that has initialised the GPU must not exec or fork
    GPU work).  Rank r uses GPU r % (devices on the box); on a one-GPU box that only works with the
    rehearsal transport (GPRN_COMM_TRANSPORT=shm)."""
    port = os.environ.get('MASTER_PORT') or str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, GPRN_LAUNCH_TAG=str(os.getpid()))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every child: as soon as one has failed the others are stopped (a rank that died before the
    # communicator exists would leave the rest waiting in the rendezvous, holding the GPUs), and the whole
    # launch has a time limit (GPRN_LAUNCH_TIMEOUT_S, default 1500 s)
    import tempfile
    import threading
    buf = tempfile.TemporaryFile()
    pump = threading.Thread(target=lambda: buf.write(procs[0].stdout.read()), daemon=True)
    pump.start()
    deadline = time.time() + float(os.environ.get('GPRN_LAUNCH_TIMEOUT_S', 1500))
    why = None
    while True:
        codes = [pr.poll() for pr in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            why = 'a rank failed'
        elif time.time() > deadline:
            why = 'time limit reached'
        if why:
            for pr in procs:
                if pr.poll() is None:
                    pr.terminate()
            t_kill = time.time() + 10
            while any(pr.poll() is None for pr in procs) and time.time() < t_kill:
                time.sleep(0.1)
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
            codes = [pr.wait() for pr in procs]
            break
        time.sleep(0.2)
    pump.join(timeout=10)
    buf.seek(0)
    sys.stdout.write(buf.read().decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or why:
        sys.exit('bench.py: %srank(s) failed: %s' % (why + '; ' if why else '',
                                                     ', '.join('%d (exit %d)' % rc for rc in bad)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-calc', action='store_true')
    ap.add_argument('--shape', default=None,
                    help='N,p,q of an ad-hoc problem (experiments; not a BASELINE config)')
    ap.add_argument('--blocks', type=int, default=20, help='timed blocks of --steps sweeps each')
    ap.add_argument('--latency', action='store_true',
                    help='the small-N regime instead: nELBO evaluations/s at N = 45, 200, 497, 512, 2048 (one JSON line each)')
    ap.add_argument('--latency-reps', type=int, default=0, help='evaluations per shape (default 200, 40 at N = 2048)')
    ap.add_argument('--latency-only', default='', help='comma-separated N of the shapes to run (default: all five)')
    ap.add_argument('--latency-cpu-s', type=float, default=20.0, help='seconds of CPU baseline per shape')
    ap.add_argument('--latency-batch', type=int, default=0, help='evaluations per side-by-side call (default 256 up to two tiles, 32 above)')
    ap.add_argument('--no-side', action='store_true', help='skip the side-by-side leg of --latency')
    ap.add_argument('--latency-mcmc', type=int, default=3,
                    help='ensemble steps of the inference.mcmc leg of --latency (both ways, N <= 512; 0: skip)')
    ap.add_argument('--also-config', default=None,
                    help='C or C:N -- after the headline, BASELINE config C (at N, for rehearsals) sharded over the same ranks, '
                         'reported under configs_at_n_gpus; default: config 4 at --gpus 4, config 5 at --gpus 8 (their stated '
                         'topologies), "none" to skip')
    a = ap.parse_args()
    if a.latency:
        return latency(a)

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        return self_launch(a)
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    a.gpus = world
    if os.environ.get('GPRN_BENCH_LAUNCH_PROBE'):
        # tests/test_sharding.py: what a rank sees, without touching a GPU
        if rank == 0:
            print(json.dumps({'probe': True, 'world': world, 'rank': rank,
                              'env': {k: os.environ.get(k) for k in ('LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                                     'GPRN_LAUNCH_TAG')}}), flush=True)
        sys.exit(int(os.environ.get('GPRN_BENCH_PROBE_FAIL_RANK', -1)) == rank)
    N, p, q, kind = synth.CONFIGS[a.config]
    if a.shape:
        N, p, q = (int(x) for x in a.shape.split(','))
    comm = sharding.Comm() if world > 1 else None

    t, ys, es = synth.rv_series(N, p)
    spec = synth.component_spec(p, q, kind)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair], comm=comm)
    g.set_components(nodes, weights, means, jit)

    # the CPU baseline FIRST (rank 0 of a one-rank run only; ~3 min at config 3): the timed GPU blocks are then the last
    # thing the command does, where a utilisation sampler watching the run can see them
    cpu = cpu_baseline(N, p, q, kind) if (world == 1 and rank == 0 and not a.no_cpu) else None

    t0 = time.time()
    ctx = g._setup_device(nodes, weights, means, jit)     # fill + chol(K): once per ELBOcalc
    ctx.barrier_max(0.0)
    t_setup = time.time() - t0
    ms_fill, n_fill, ms_fill_pass = 0.0, 0, 0.0
    mu0, var0 = g._initMuVar(nodes, weights, jit)
    ctx.set_muvar(mu0, var0)

    # secondary metric: full ELBOcalc with changed hyper-parameters, i.e. fused
    # fills + chol(K) + inverses + the reference's own trip count (discarded sweep + loop to the
    # stop rule), allocations warm.  Same control flow on every rank (the ELBO is all-reduced).
    calc = None
    if not a.no_calc:
        times, trips = [], []
        for rep in range(2):
            for j, node in enumerate(nodes):
                node.pars[1] *= 1.0 + 1e-3 * (rep + 1)          # new length scale -> refill + refactor
            ctx.barrier_max(0.0)
            t0 = time.perf_counter()
            _, _, _, it = g.ELBOcalc()
            times.append(ctx.barrier_max(time.perf_counter() - t0))
            trips.append(it)
        # the covariance fills of one more set-up, timed per launch (HIP events; code is warm here)
        for j, node in enumerate(nodes):
            node.pars[1] *= 1.0 + 1e-3
        ctx.profile_enable(['fill'])
        g._setup_device(nodes, weights, means, jit)
        ms_fill, n_fill = ctx.profile_read()['fill']
        ctx.profile_enable([])
        ms_fill_pass = ctx.fill_rate(20)                        # the same fills, launch behind launch, in ONE event bracket
        calc = {'elbocalc_per_s': 1.0 / min(times), 'ms': 1e3 * min(times), 'loop_trips': trips[-1]}
        if world == 1:
            # GPRN prediction at 1000 new times from the converged state
            # (8 x {fill, chol + inverse, K* fill, X K*^T} on the device + the O(pqN*) mix on the host)
            g.predict(nn=1000)
            t0 = time.perf_counter()
            g.predict(nn=1000)
            calc['predict_1000_ms'] = 1e3 * (time.perf_counter() - t0)

    # SURVEY.md 8f-1, the other way to use N GPUs: independent ELBO evaluations (optimiser
    # populations, emcee walkers), every rank with the whole problem on its own GPU, no
    # exchange -- aggregate full ELBOcalc evaluations per second over all ranks (weak scaling).
    pool = None
    if world > 1 and not a.no_calc:
        local_dev = comm.local_rank % max(1, _hip.device_count())
        g1 = gpyrn.inference(q, t, *[x for pair in zip(ys, es) for x in pair], device=local_dev)
        n1, w1, m1, j1 = synth.build_components(covfunc, meanfunc, spec)
        g1.set_components(n1, w1, m1, j1)
        g1.ELBOcalc()                                            # allocations, code, first factors
        reps = 2
        ctx.barrier_max(0.0)
        t0 = time.perf_counter()
        for rep in range(reps):
            for node in n1:
                node.pars[1] *= 1.0 + 1e-3 * (rep + 1 + rank)   # every evaluation refills and refactors
            g1.ELBOcalc()
        dt_pool = ctx.barrier_max(time.perf_counter() - t0)
        pool = {'elbocalc_per_s_all_ranks': world * reps / dt_pool, 'ms_per_elbocalc': 1e3 * dt_pool / reps,
                'scaling': 'weak', 'note': 'one independent full ELBOcalc stream per rank'}

    # ---- the headline: timed blocks of forced sweeps, last (the secondary metrics above changed the hyper-parameters:
    # back to the configuration's own)
    nodes, weights, means, jit = synth.build_components(covfunc, meanfunc, spec)
    g.set_components(nodes, weights, means, jit)
    ctx = g._setup_device(nodes, weights, means, jit)
    ctx.set_muvar(mu0, var0)
    golden = golden_check(ctx, a, N, p, q, t, ys) if world == 1 else None
    if a.warmup > 0:
        ctx.sweep(a.warmup, commit=True)
    ctx.profile_enable(['update', 'update_ahead'])
    block_s = []
    for _ in range(max(1, a.blocks)):
        ctx.barrier_max(0.0)                               # barrier + device sync
        t0 = time.perf_counter()
        elbo, parts, info = ctx.sweep(a.steps, commit=True)    # K sweeps, one host sync at the end
        dt_local = time.perf_counter() - t0
        block_s.append(ctx.barrier_max(dt_local))          # MAX over ranks
    dt = float(np.median(block_s))
    prof = ctx.profile_read()
    ctx.profile_enable([])

    # BASELINE's multi-GPU configs in their stated topologies (config 4: 4 GPUs, config 5: 8 GPUs) ride along with the
    # config-3 strong-scaling headline of a run that has exactly that many ranks
    extra = {}
    also = a.also_config if a.also_config is not None else {4: '4', 8: '5'}.get(world, 'none')
    if world > 1 and also != 'none' and not a.shape:
        cfg_s, _, n_s = also.partition(':')
        blk = config_in_topology(int(cfg_s), world, rank, n_override=int(n_s) if n_s else None)
        if rank == 0:
            extra['config %s on %d ranks' % (cfg_s, world)] = blk

    if rank == 0:
        nodes_l, weights_l = sharding.local_gps(p, q, world, 0)
        ms_upd, n_upd = prof['update']
        ms_ahd, n_ahd = prof.get('update_ahead', (0.0, 0))
        T = (N + TILE - 1) // TILE
        split = T <= 64                                             # csrc/factor.hip: "rest" in two launches up to 64 tile steps
        per_sweep = update_flops(T, len(nodes_l) + len(weights_l), split=split)
        fl, fl_ahd = (x * a.steps * len(block_s) for x in per_sweep[:2])
        uni = k512_union() if world == 1 and a.config == 3 and not a.shape else None
        achieved = fl / (ms_upd * 1e-3) / 1e12 if ms_upd > 0 else None
        out = {
            'metric': 'ELBO iterations/sec (N=%d, P=%d, Q=%d)' % (N, p, q),
            'value': a.steps / dt,
            'unit': 'sweeps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * dt / a.steps,
            'blocks': {'n': len(block_s), 'steps_each': a.steps,
                       'sweeps_per_s': {'median': a.steps / dt, 'min': a.steps / max(block_s),
                                        'max': a.steps / min(block_s)}},
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '%s: N=%d, p=%d outputs, q=%d nodes, %s nodes / SE '
                                   'weights, synthetic RV series (seed 0)'
                                   % ('ad-hoc shape' if a.shape else 'BASELINE config %d' % a.config,
                                      N, p, q, kind),
                       'latent_gps': q * (p + 1), 'sharding': 'latent GPs over %d rank(s)' % world,
                       'comm_ranks': ctx.world, 'transport': os.environ.get('GPRN_COMM_TRANSPORT', 'rccl') if world > 1 else None},
            'sweep_tflops': sweep_flops(N, p, q) * a.steps / dt / 1e12,
            # BASELINE metric, second half: fp64 rate of the Cholesky work.  The sweep's N^3 work IS
            # the G fused Cholesky + triangular-inverse factorisations (+ q-1 X^T X products); this is
            # that flop count over the whole sweep time, O(N^2) kernels and exchanges included.
            'cholesky_gflops': sweep_flops(N, p, q) * a.steps / dt / 1e9,
            # fused covariance fill at setup: 8 N^2 bytes written per matrix, against the HBM peak
            # (it is fp64-VALU-bound, not HBM-bound: a division and exp/sin per element)
            # (kernel_GBps: the launches back to back inside one event bracket -- the rate the kernels run at; GBps: every
            # launch bracketed by its own pair of events, gaps included)
            'fill': ({'kernel_GBps': (n_fill * 8.0 * N * N / (ms_fill_pass * 1e-3) / 1e9) if ms_fill_pass > 0 else None,
                      'kernel_frac': (n_fill * 8.0 * N * N / (ms_fill_pass * 1e-3) / 1e9 / HBM_PEAK_GBPS) if ms_fill_pass > 0 else None,
                      'GBps': n_fill * 8.0 * N * N / (ms_fill * 1e-3) / 1e9, 'launches': n_fill,
                      'peak_GBps': HBM_PEAK_GBPS,
                      'frac': n_fill * 8.0 * N * N / (ms_fill * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                     if ms_fill > 0 else None),
            'setup_s': t_setup,
            'full_elbocalc': calc,
            'independent_evaluations': pool,
            # the first sweeps against the reference's recorded values for this workload (golden_check), then the ELBO the
            # timed blocks ended on
            'golden_check': golden,
            'elbo_last': float(elbo[-1]), 'info': int(info),
            # which schedule produced the line: 1 = device-side flags (the default), 0 = HIP events; fallbacks = calls of this
            # context that were re-run on events after an in-kernel wait timed out (0 in a healthy run)
            'schedule': {'flags': int(ctx.option('flags')), 'fallbacks': int(ctx.option('fallbacks'))},
            'roofline': {
                'kernel': 'k_tile_gemm<..., TG_BULK> (bulk trailing-update launches, K=512, v_mfma_f64_16x16x4_f64)',
                'bound': 'mfma', 'achieved': achieved, 'peak': FP64_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': (achieved / FP64_MFMA_PEAK_TFLOPS) if achieved else None,
                'traffic': pmc_traffic() if world == 1 and a.config == 3 and not a.shape else None,
                'pmc': pmc_mfma() if world == 1 and a.config == 3 and not a.shape else None,
                'measured_mfma_ceiling': ctx.mfma_peak(2, 4000),
                # the whole sweep against the same peak (every kernel, wait and O(N^2) step included)
                'sweep_frac': sweep_flops(N, p, q) * a.steps / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                # rate of ALL K = 512 launches (next-panel + look-ahead + bulk) over the time at least one of them was
                # open, from the committed kernel trace: a launch's own rate divides by the time it is open, during which
                # up to three other streams' kernels share the CUs with it
                'k512_union': ({'tflops': sum(per_sweep) / (uni['k512_union_us_per_sweep'] * 1e-6) / 1e12,
                                'union_us_per_sweep': uni['k512_union_us_per_sweep'],
                                'bulk_ahead_tflops': (per_sweep[0] + per_sweep[1]) / (uni['bulk_ahead_union_us_per_sweep'] * 1e-6) / 1e12,
                                'source': uni.get('source')} if uni else None),
                'launches': n_upd, 'avg_launch_ms': (ms_upd / n_upd) if n_upd else None,
                'flops_per_launch': (fl / n_upd) if n_upd else None,
                # the look-ahead part of the same updates (own launches, k_tile_gemm<..., TG_AHEAD>: the tiles the
                # next panel's update writes again; they run beside that panel's "next" launch)
                'ahead_launches': ({'launches': n_ahd, 'avg_launch_ms': ms_ahd / n_ahd, 'flops_per_launch': fl_ahd / n_ahd,
                                    'achieved': fl_ahd / (ms_ahd * 1e-3) / 1e12} if n_ahd else None),
            },
        }
        out['configs_at_n_gpus'] = extra or None
        out['cpu_baseline'] = cpu
        print(json.dumps(out), flush=True)
    if comm is not None:
        ctx.barrier_max(0.0)
        comm.cleanup()


if __name__ == '__main__':
    main()
