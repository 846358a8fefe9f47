"""One rank of a sharded run on a golden case (started by test_parity_gpu.py).

usage: python -m tests._shard_worker <tag> <out.npz> <rendezvous tag>
with RANK / WORLD_SIZE / LOCAL_RANK in the environment.
"""
import os
import sys

import numpy as np

import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc, sharding
from tests import _cases


def main(tag, out, uid_tag):
    meta, d = _cases.load(tag)
    nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
    comm = sharding.Comm(tag=uid_tag)
    g = gpyrn.inference(meta['q'], np.array(d['time']), *_cases.data_args(d), comm=comm)
    if os.environ.get('GPRN_TEST_USER_WEIGHTS'):
        # the weights as user-defined covFunction subclasses: host-evaluated matrices (gprn_upload_K, and
        # gprn_predict_upload in prediction) on whichever rank owns them
        class UserKernel(covfunc.covFunction):
            def __init__(self, inner):
                super().__init__(*inner.pars)
                self._inner = inner
                self._param_names = inner._param_names

            def __call__(self, r):
                return self._inner(r)
        weights = [UserKernel(w) for w in weights]
        assert weights[0]._device_program() is None
    g.set_components(nodes, weights, means, jit)
    res = {}
    # forced sweeps from the reference's own initial state
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    bad_rank = int(os.environ.get('GPRN_TEST_BAD_RANK', -1))
    if bad_rank >= 0:
        # ONE rank fails a local check (here: it never set the state).  The ranks agree on their checks before
        # anybody issues a collective, so every rank returns -- the one with the finding its own error, the others "another
        # rank did not pass its checks" -- instead of the others waiting in a broadcast for ever
        from gpyrn_amd import _hip
        if comm.rank != bad_rank:
            ctx.set_muvar(d['mu_init'], d['var_init'])
        try:
            ctx.sweep(1, commit=True)
            msg = 'no error'
        except _hip.BackendError as exc:
            msg = str(exc)
        np.savez(out, rank=comm.rank, world=comm.world, message=msg)
        comm.cleanup()
        return
    ctx.set_muvar(d['mu_init'], d['var_init'])
    if os.environ.get('GPRN_TEST_LONG_SWEEPS'):
        # the watchdog test: sweep until killed.  A marker file says that this rank is past its first call (the harness kills
        # one rank after both markers exist: the other is then inside a sweep's collectives, or about to enter them)
        ctx.sweep(1, commit=True)
        open(out + '.started', 'w').close()
        for _ in range(int(os.environ['GPRN_TEST_LONG_SWEEPS'])):
            ctx.sweep(5, commit=False)
        np.savez(out, rank=comm.rank, world=comm.world, message='ran to the end')
        comm.cleanup()
        return
    if os.environ.get('GPRN_TEST_LONG_ELBOCALC'):
        # ONE gprn_elbocalc call that legitimately outlasts the watchdog's budget (q = 3: the reference's iteration
        # diverges, the stop rule never fires, the loop runs to max_iter) -- the budget bounds a stall, not a call's length
        import time
        t0 = time.time()
        E, mu, var, it = g.ELBOcalc(max_iter=int(os.environ['GPRN_TEST_LONG_ELBOCALC']))
        np.savez(out, rank=comm.rank, world=comm.world, seconds=time.time() - t0, iters=it)
        comm.cleanup()
        return
    hook_rank = int(os.environ.get('GPRN_TEST_WITHHOLD_RANK', -1))
    if hook_rank == comm.rank:
        # this rank only: a producer flag that never goes up, so that its in-kernel wait gives up after 20 ms
        ctx.option('wait_budget_ms', 20)
        ctx.option('withhold_inner', 2)
    elbo, parts, info = ctx.sweep(meta['nsweeps'], commit=True)
    if hook_rank == comm.rank:
        ctx.option('withhold_inner', 0)
        ctx.option('wait_budget_ms', 2000)
    mu, var = ctx.get_muvar()
    res.update(sw_elbo=elbo, sw_parts=parts, sw_info=info, sw_mu=mu, sw_var=var,
               logdet_K=ctx.get_logdet_K(), fallbacks=ctx.option('fallbacks'), flags=ctx.option('flags'))
    if os.environ.get('GPRN_TEST_PREDICT'):
        ref = np.load(os.path.join(_cases.GOLDEN, 'pred_' + tag + '.npz'))
        pm, pv, parts_ = g._Prediction(tstar=ref['tstar'], mu=d['mu_final'], var=d['var_final'], separate=True)
        res.update(pred_mean=pm, pred_var=pv, pred_nodes=np.array(parts_[0], dtype=float),
                   pred_weights=np.array(parts_[1], dtype=float), pred_info=g.last_info)
    if 'calc_elbo' in d:
        E, mu, var, it = g.ELBOcalc()
        res.update(calc_elbo=E, calc_mu=mu, calc_var=var, calc_iter=it,
                   calc_history=np.array(g._elbo_history))
    np.savez(out, rank=comm.rank, world=comm.world, **res)
    comm.cleanup()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
