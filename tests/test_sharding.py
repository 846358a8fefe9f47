"""Multi-rank path on CPU (gloo, world_size 2): the sharding schedule of
csrc/api.hip / api_sweep.hip -- who owns which latent GP, what is broadcast after each
half-sweep, which scalars are all-reduced -- executed with the oracle's per-GP
arithmetic in place of the HIP kernels, and compared with the unsharded sweep."""
import os
import socket

import numpy as np
import pytest

from gpyrn_amd import covfunc, meanfunc, sharding
from oracle import cpu_ref
from tests import _cases


def test_owner_map_covers_and_balances():
    for p, q, world in [(3, 2, 1), (3, 2, 2), (3, 4, 4), (4, 3, 8), (1, 1, 4), (3, 2, 8)]:
        own = sharding.owners(p, q, world)
        assert len(own) == q * (p + 1) and all(0 <= o < world for o in own)
        seen = []
        for r in range(world):
            n, w = sharding.local_gps(p, q, world, r)
            seen += n + w
            assert len(n) <= -(-q // world) and len(w) <= -(-q * p // world)
        assert sorted(seen) == list(range(q * (p + 1)))
    # BASELINE config 4: one node and three weights per GPU
    assert [len(sharding.local_gps(3, 4, 4, r)[0]) for r in range(4)] == [1, 1, 1, 1]
    assert [len(sharding.local_gps(3, 4, 4, r)[1]) for r in range(4)] == [3, 3, 3, 3]


def test_helper_inverses():
    assert sharding.helper_inverses(3, 1, 2, 0) == []
    assert sharding.helper_inverses(3, 2, 1, 0) == [1]
    assert sharding.helper_inverses(3, 4, 4, 0) == [1, 2, 3]
    assert sharding.helper_inverses(3, 4, 4, 3) == []
    assert sharding.helper_inverses(3, 4, 8, 5) == []          # owns no node


def _row(g, p, q):
    if g < q:
        return g
    j, i = divmod(g - q, p)
    return (1 + i) * q + j


def sharded_sweep(rank, world, dist, torch, Kf, Kw, Lf, Lw, y, y_raw, yerr2, jitt2, mu, var):
    """One sweep with the latent GPs sharded over `world` ranks (schedule of
    gprn_sweep in csrc/api_sweep.hip; arithmetic of oracle/cpu_ref.sweep_B)."""
    q, N = Kf.shape[0], Kf.shape[-1]
    p = Kw.shape[0] // q
    G = q * (p + 1)
    own = sharding.owners(p, q, world)
    nodes_l, weights_l = sharding.local_gps(p, q, world, rank)
    Kinv = {j: cpu_ref.cho_solve((Lf[j], True), np.eye(N))
            for j in sharding.helper_inverses(p, q, world, rank)}
    variance = jitt2[:, None] + yerr2
    muF, muW = cpu_ref.split_u(mu, p, q, N)
    varF, varW = cpu_ref.split_u(var, p, q, N)
    state_mu = np.concatenate((muF[None], muW)).reshape((p + 1) * q, N).copy()
    state_var = np.concatenate((varF[None], varW)).reshape((p + 1) * q, N).copy()
    scal = np.zeros(3 * G + q * q)                   # logdetB, trBinv, muKmu, q1
    logdetK = np.array([2 * np.sum(np.log(np.diag(L))) for L in list(Lf) + list(Lw)])

    def exchange(gps):
        for g in gps:
            for st in (state_mu, state_var):
                t = torch.from_numpy(st[_row(g, p, q)])
                dist.broadcast(t, src=own[g])

    # ---- node half-sweep (Jacobi: every local node sees the old state)
    new_rows = {}
    for j in nodes_l:
        d, pred = cpu_ref._node_d_and_pred(y, variance, muF, muW, varW, j)
        ds, m, ldB, trB, Binv, s = cpu_ref._gp_update_B(Kf[j], d, pred, need_inverse=(j < q - 1))
        new_rows[j] = (m, ds)
        scal[j], scal[G + j] = ldB, trB
        for jj in range(j + 1, q):                   # quirk Q1 on the owner of the earlier node
            Sk = (np.eye(N) - Binv) / (s[:, None] * s[None, :])
            scal[3 * G + jj * q + j] = np.sum(Kinv[jj] * Sk)
    for j, (m, ds) in new_rows.items():
        state_mu[j], state_var[j] = m, ds
    exchange(range(q))
    mu_f, dsf = state_mu[:q].copy(), state_var[:q].copy()

    # ---- weight half-sweep (new nodes, old weights)
    new_rows = {}
    for g in weights_l:
        j, i = divmod(g - q, p)
        d, pred = cpu_ref._weight_d_and_pred(y, variance, mu_f, dsf, muW, j, i)
        ds, m, ldB, trB, _, _ = cpu_ref._gp_update_B(Kw[g - q], d, pred)
        new_rows[g] = (m, ds)
        scal[g], scal[G + g] = ldB, trB
    for g, (m, ds) in new_rows.items():
        state_mu[_row(g, p, q)], state_var[_row(g, p, q)] = m, ds
    exchange(range(q, G))

    # ---- mu^T K^-1 mu with the state row g (quirk Q2 for the weights), then one all-reduce
    for g in nodes_l + weights_l:
        L = Lf[g] if g < q else Lw[g - q]
        a = cpu_ref.solve_triangular(L, state_mu[g], lower=True)
        scal[2 * G + g] = a @ a
    t = torch.from_numpy(scal)
    dist.all_reduce(t)

    new_mu = state_mu.reshape(p + 1, q, N)
    new_var = state_var.reshape(p + 1, q, N)
    dsw = np.transpose(new_var[1:], (1, 0, 2))
    logl = cpu_ref.expected_loglike(y_raw, variance, new_mu[0], new_mu[1:], new_var[0], dsw)
    ent = 0.5 * np.sum(logdetK - scal[:G]) + 0.5 * q * (p + 1) * N * (1 + cpu_ref.LOG2PI)
    logp = -0.5 * N * q * (p + 1) * cpu_ref.LOG2PI
    for g in range(G):
        tr = scal[G + g]
        if g < q:
            tr += sum(scal[3 * G + g * q + k] for k in range(g))
        logp += -0.5 * logdetK[g] - 0.5 * (scal[2 * G + g] + tr)
    return (logl + logp + ent) / q, new_mu, new_var


def _worker(rank, world, port, tag, out_dir):
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank,
                            world_size=world)
    try:
        meta, d = _cases.load(tag)
        nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
        Kf, Kw, Lf, Lw, y, j2 = cpu_ref.setup(d['time'], nodes, weights, means, jit, d['y'])
        mu, var = d['mu_init'], d['var_init']
        elbos = []
        for _ in range(2):
            E, mu, var = sharded_sweep(rank, world, dist, torch, Kf, Kw, Lf, Lw, y, d['y'],
                                       d['yerr']**2, j2, mu, var)
            elbos.append(E)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), elbo=np.array(elbos), mu=mu, var=var)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('tag', ['step_p3q2', 'step_p2q3'])
def test_two_ranks_gloo_match_unsharded(tag, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), tag, str(tmp_path)), nprocs=world, join=True)
    meta, d = _cases.load(tag)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    assert np.array_equal(r0['elbo'], r1['elbo']) and np.array_equal(r0['mu'], r1['mu'])
    # against the reference's golden sweeps and the unsharded oracle
    np.testing.assert_allclose(r0['elbo'], d['elbo_sweeps'][:2], rtol=1e-9)
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    args = cpu_ref.setup(d['time'], nodes, weights, means, jit, d['y'])
    mu, var = d['mu_init'], d['var_init']
    for s in range(2):
        E, mu, var, _ = cpu_ref.sweep_B(*args[:5], d['y'], d['yerr']**2, args[5], mu, var)
        np.testing.assert_allclose(r0['elbo'][s], E, rtol=1e-12)
    np.testing.assert_allclose(r0['mu'], mu, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r0['var'], var, rtol=1e-10, atol=1e-14)


def test_eight_ranks_gloo_config5_partition(tmp_path):
    """BASELINE config 5's stated topology -- 15 latent GPs (q = 3 nodes, p = 4 outputs) over EIGHT ranks -- which no GPU box
    of the build can host (its process guard admits six processes on the card; tests/test_parity_gpu.py runs the library
    itself at five and bench.py's shm rehearsal at six, profiles/r06_bench_6ranks_shm_one_gpu.json): the partition that the
    library executes (sharding.owners / local_gps / helper_inverses are what csrc/api.hip is fed) with eight gloo processes and
    the oracle's per-GP arithmetic, at the size the reference itself was run (cfg5shape_N1024).  Five ranks own no node,
    seven ranks own two latent GPs and one a single weight, only ranks 0 and 1 refactor a later node's K_j (quirk Q1); every rank must end
    with the reference's two sweeps."""
    import torch.multiprocessing as mp
    tag, world = 'cfg5shape_N1024', 8
    meta, d = _cases.load(tag)
    p, q = meta['p'], meta['q']
    assert (p, q) == (4, 3)
    parts = [sharding.local_gps(p, q, world, r) for r in range(world)]
    assert [len(n) for n, _ in parts] == [1, 1, 1, 0, 0, 0, 0, 0]
    assert [len(n) + len(w) for n, w in parts] == [2, 2, 2, 2, 2, 2, 2, 1]
    assert [bool(sharding.helper_inverses(p, q, world, r)) for r in range(world)] == [True, True] + [False] * 6
    os.environ['OMP_NUM_THREADS'] = os.environ['OPENBLAS_NUM_THREADS'] = '1'     # (eight processes on the container's eight cores)
    try:
        mp.spawn(_worker, args=(world, _free_port(), tag, str(tmp_path)), nprocs=world, join=True)
    finally:
        os.environ.pop('OMP_NUM_THREADS', None); os.environ.pop('OPENBLAS_NUM_THREADS', None)
    res = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    for r in res[1:]:
        assert np.array_equal(res[0]['elbo'], r['elbo']) and np.array_equal(res[0]['mu'], r['mu'])
    np.testing.assert_allclose(res[0]['elbo'], d['elbo_sweeps'][:2], rtol=1e-8)
    _cases.assert_state('eight gloo ranks, ' + tag, res[0]['mu'], d['mu_final'], res[0]['var'], d['var_final'])


def test_eval_pool_map_logic_without_a_gpu(monkeypatch):
    """EvalPool.map: rank r evaluates items r::world, a sum over ranks with zeros elsewhere
    rebuilds the full list (scalars, tuples, -inf, nan).  The library context is replaced by a
    recorder; the two ranks run one after the other and their buffers are summed by hand."""
    from gpyrn_amd import _hip, sharding

    class FakeCtx:
        def __init__(self, device):
            self.sent = None

        def comm_init(self, world, rank, uid):
            pass

        max = None

        def barrier_max(self, v=0.0):
            return float(v) if FakeCtx.max is None else FakeCtx.max

        def allreduce_sum(self, buf):
            self.sent = np.array(buf, dtype=float)
            return FakeCtx.total if FakeCtx.total is not None else self.sent

    monkeypatch.setattr(_hip, 'Context', FakeCtx)
    monkeypatch.setattr(_hip, 'device_count', lambda: 1)

    class NoRendezvous(sharding.Comm):
        def unique_id(self, timeout=0):
            return b'\0' * 128

    f = lambda i: (float(i), -np.inf if i == 1 else (np.nan if i == 2 else 0.5 * i))
    FakeCtx.total = None
    pools = [sharding.EvalPool(NoRendezvous(world=2, rank=r, local_rank=r)) for r in range(2)]
    calls = [[], []]
    for r, pool in enumerate(pools):                 # first pass: record what each rank contributes
        pool.map(lambda i, r=r: (calls[r].append(i), f(i))[1], range(5))
    assert calls == [[0, 2, 4], [1, 3]]
    with np.errstate(invalid='ignore'):
        FakeCtx.total = pools[0]._ctx.sent + pools[1]._ctx.sent
    for pool in pools:                               # second pass: the summed buffer comes back
        out = pool.map(f, range(5))
        assert [o[0] for o in out] == [0.0, 1.0, 2.0, 3.0, 4.0]
        assert np.isneginf(out[1][1]) and np.isnan(out[2][1]) and out[4][1] == 2.0
    one = sharding.EvalPool(NoRendezvous(world=1, rank=0, local_rank=0))
    assert one.map(lambda x: x * 2.0, [1.0, 2.0]) == [2.0, 4.0]
    # map_lists: a rank's whole share goes to func as ONE list (nELBO_batch: side by side on that rank's GPU)
    FakeCtx.total = None
    shares = [[], []]
    for r, pool in enumerate(pools):
        pool.map_lists(lambda xs, r=r: (shares[r].extend(xs), [10.0 * x for x in xs])[1], range(5))
    assert shares == [[0, 2, 4], [1, 3]]
    FakeCtx.total = pools[0]._ctx.sent + pools[1]._ctx.sent
    for pool in pools:
        assert pool.map_lists(lambda xs: [10.0 * x for x in xs], range(5)) == [0.0, 10.0, 20.0, 30.0, 40.0]
    assert one.map_lists(lambda xs: [x + 1.0 for x in xs], [1.0, 2.0]) == [2.0, 3.0]
    with pytest.raises(ValueError):
        one.map_lists(lambda xs: [0.0], [1.0, 2.0])
    # (a rank without a share does not call func at all)
    FakeCtx.total = np.zeros(1)
    assert pools[1].map_lists(lambda xs: [7.0], [3.0]) == [0.0]
    # take_from_highest: the arrays of the rank with the largest key; every rank contributes zeros otherwise
    a0, a1 = np.full((2, 3), 1.5), np.full((2, 3), 2.5)
    FakeCtx.total = None
    FakeCtx.max = 4.0
    pools[0].take_from_highest(2, [a0]); sent0 = pools[0]._ctx.sent
    pools[1].take_from_highest(4, [a1]); sent1 = pools[1]._ctx.sent
    assert not sent0.any() and np.array_equal(sent1, a1.ravel())
    FakeCtx.total = sent0 + sent1
    for pool, key, arr in ((pools[0], 2, a0), (pools[1], 4, a1)):
        got = pool.take_from_highest(key, [arr])
        assert got[0].shape == (2, 3) and np.array_equal(got[0], a1)
    FakeCtx.max = -1.0
    assert pools[0].take_from_highest(-1, [a0]) is None
    FakeCtx.max = None
    assert np.array_equal(one.take_from_highest(0, [a0])[0], a0) and one.take_from_highest(-1, [a0]) is None


def test_pooled_side_by_side_evaluations_keep_one_state_on_every_rank():
    """inference._nELBO_batch_pool without a GPU: three 'ranks' (threads, each with its own inference object and an
    EvalPool over an in-memory all-reduce) split seven vectors; the device call is replaced by a recorder that returns
    sum(x) and 'converges' on chosen vectors.  Every rank must return the full list in order, and every rank must end
    with the state of the LAST vector of the WHOLE list that converged -- whoever evaluated it -- or keep its state
    when none did."""
    import threading
    import gpyrn_amd as gpyrn
    world = 3

    class Wire:
        def __init__(self):
            self.bar = threading.Barrier(world)
            self.slots = [None] * world

        def reduce(self, rank, buf, op):
            self.slots[rank] = np.array(buf, dtype=float)
            self.bar.wait()
            out = op(np.stack(self.slots), axis=0)
            self.bar.wait()
            return out

    class Ctx:
        def __init__(self, wire, rank):
            self.wire, self.rank = wire, rank

        def barrier_max(self, v=0.0):
            return float(self.wire.reduce(self.rank, [v], np.max)[0])

        def allreduce_sum(self, buf):
            return self.wire.reduce(self.rank, np.ravel(buf), np.sum)

    def pool_for(wire, rank):
        pool = sharding.EvalPool.__new__(sharding.EvalPool)
        pool.world, pool.rank, pool._ctx = world, rank, Ctx(wire, rank)
        return pool

    t = np.linspace(0.0, 10.0, 12)
    sets = [np.array([1.0 + 0.1 * k, 2.0, 0.5, 3.0, 0.0, 0.1]) for k in range(7)]

    def run(converging, one_by_one=False, stored=True):
        wire, out = Wire(), [None] * world

        def rank_main(r):
            g = gpyrn.inference(1, t, np.sin(t), 0.1 * np.ones(t.size))
            g.set_components(covfunc.SquaredExponential(1.0, 2.0), covfunc.SquaredExponential(0.5, 3.0),
                             meanfunc.Constant(0.0), 0.1)
            if stored:
                g._mu = np.full((2, 1, t.size), -1.0)
                g._var = np.full((2, 1, t.size), -2.0)
            seen = []

            def fake_nelbo(x, max_iter=None):                  # what nELBO does to the object: a state only when it converged
                seen.append(float(x[0]))
                g.set_parameters(x)
                if round((x[0] - 1.0) * 10) in converging:
                    g._mu = np.full((2, 1, t.size), x[0])
                    g._var = np.full((2, 1, t.size), 10.0 * x[0])
                return float(np.sum(x))

            def fake_device(xs, max_iter):                     # what _nELBO_batch_device does to the object
                seen.extend(float(x[0]) for x in xs)
                done = [i for i, x in enumerate(xs) if round((x[0] - 1.0) * 10) in converging]
                g._batch_last_done = done[-1] if done else -1
                if done:
                    g._mu = np.full((2, 1, t.size), xs[done[-1]][0])
                    g._var = np.full((2, 1, t.size), 10.0 * xs[done[-1]][0])
                return [float(np.sum(x)) for x in xs]

            g._nELBO_batch_device = (lambda xs, max_iter: None) if one_by_one else fake_device
            g.nELBO = fake_nelbo
            g._batchable = lambda: True
            vals = g.nELBO_batch(sets, pool=pool_for(wire, r))
            out[r] = (vals, None if g._mu is None else g._mu.copy(), None if g._var is None else g._var.copy(), seen,
                      np.array(g.get_parameters()))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(60)
        assert all(o is not None for o in out), 'a rank did not finish'
        return out

    out = run({0, 2, 4})                                       # the last converged one, vector 4, is rank 1's
    for r, (vals, mu, var, seen, pars) in enumerate(out):
        np.testing.assert_allclose(vals, [float(np.sum(x)) for x in sets], rtol=1e-15)
        assert seen == [sets[i][0] for i in range(r, 7, world)]
        assert np.all(mu == sets[4][0]) and np.all(var == 10.0 * sets[4][0])
        np.testing.assert_array_equal(pars, sets[-1])
    out = run(set())                                           # nothing converged: every rank keeps the state it had
    for vals, mu, var, seen, pars in out:
        assert np.all(mu == -1.0) and np.all(var == -2.0)
    # a share that cannot run side by side is evaluated one by one.  The state handed round is again that of the
    # last evaluation of the whole list that CONVERGED (vector 3: rank 0's second, not its last), and a rank on which
    # nothing converged and nothing was stored (no state at all) offers nothing -- the collective's buffers stay the same
    # size on every rank
    out = run({1, 3}, one_by_one=True, stored=False)
    for r, (vals, mu, var, seen, pars) in enumerate(out):
        np.testing.assert_allclose(vals, [float(np.sum(x)) for x in sets], rtol=1e-15)
        assert seen == [sets[i][0] for i in range(r, 7, world)]
        assert np.all(mu == sets[3][0]) and np.all(var == 10.0 * sets[3][0])
    out = run(set(), one_by_one=True, stored=False)            # nothing converged, nothing stored: still nothing
    for vals, mu, var, seen, pars in out:
        assert mu is None and var is None


# ---------------------------------------------------------------- bench.py --gpus N without a launcher
def _bench(args, extra_env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=env, cwd=root,
                          capture_output=True, text=True, timeout=120)


def test_bench_self_launch_starts_one_rank_per_gpu():
    """`python bench.py --gpus 3` with no launcher around it: three child ranks, before any GPU call, and
    exactly rank 0's line on stdout (the children answer from a probe hook, no GPU needed)."""
    import json
    r = _bench(['--gpus', '3'], {'GPRN_BENCH_LAUNCH_PROBE': '1'})
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['world'] == 3 and d['rank'] == 0 and d['env']['LOCAL_RANK'] == '0'
    assert d['env']['MASTER_ADDR'] == '127.0.0.1' and d['env']['MASTER_PORT'] and d['env']['GPRN_LAUNCH_TAG']


def test_bench_self_launch_reports_a_failed_rank():
    r = _bench(['--gpus', '2'], {'GPRN_BENCH_LAUNCH_PROBE': '1', 'GPRN_BENCH_PROBE_FAIL_RANK': '1'})
    assert r.returncode != 0 and 'rank(s) failed: 1' in r.stderr


def test_bench_respects_an_outer_launcher():
    """With WORLD_SIZE set by a launcher (torch.distributed.run), bench.py is one rank and starts nothing."""
    import json
    r = _bench(['--gpus', '4'], {'GPRN_BENCH_LAUNCH_PROBE': '1', 'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip())
    assert d['world'] == 4 and d['env']['GPRN_LAUNCH_TAG'] is None
