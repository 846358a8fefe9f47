"""One rank of an EvalPool run (started by test_parity_gpu.py).

usage: python -m tests._pool_worker <tag> <out.npz> <rendezvous tag>
"""
import sys

import numpy as np

import gpyrn_amd as gpyrn
from gpyrn_amd import covfunc, meanfunc, sharding
from tests import _cases


def main(tag, out, uid_tag):
    meta, d = _cases.load(tag)
    pool = sharding.EvalPool(sharding.Comm(tag=uid_tag))

    def fresh():
        nodes, weights, means, jit = _cases.components(meta, covfunc, meanfunc)
        g = gpyrn.inference(meta['q'], np.array(d['time']), *_cases.data_args(d), device=pool.device)
        g.set_components(nodes, weights, means, jit)
        return g

    base = fresh().get_parameters()
    sets = [base * (1.0 + 0.01 * k) for k in range(5)]
    f = lambda x: float(fresh().nELBO(x))          # no warm start: independent of who evaluated what before
    pooled = pool.map(f, sets)
    serial = [f(x) for x in sets]
    # one object through the public entry point: each rank's share SIDE BY SIDE on its GPU, the state handed round --
    # cold (every vector from its own _initMuVar state: = serial), then warm from the state the first call left; the same
    # two calls without a pool must give the same values and leave the same state, whatever the number of ranks
    g1 = fresh()
    batch = g1.nELBO_batch(sets, pool=pool)
    batch_warm = g1.nELBO_batch(sets[::-1], pool=pool)
    g2 = fresh()
    alone = g2.nELBO_batch(sets)
    alone_warm = g2.nELBO_batch(sets[::-1])
    one_by_one = fresh().nELBO_batch(sets, pool=pool, batch=False)   # (the chained form: rank r's share, each from its predecessor)
    # inference.mcmc with both: emcee's vectorised log-probability, a half-step's walkers split over the ranks
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fake_emcee'))
    from scipy import stats
    names = list(np.array(list(g1.parameters_dict.keys()))[~g1.frozen_mask])
    priors = {n: stats.uniform(0.7 * abs(v) - 1e-3, 0.6 * abs(v) + 2e-3) for n, v in zip(names, base)}
    chains = []
    for pl in (pool, None):
        np.random.seed(11)
        s = fresh().mcmc(priors, niter=2, batch=True, **({'pool': pl} if pl is not None else {}))
        chains.append((s.get_chain(), s.get_log_prob()))
    # tuples, -inf and nan travel unchanged
    odd = pool.map(lambda i: (float(i), -np.inf if i == 1 else (np.nan if i == 2 else 0.5 * i)), range(7))
    np.savez(out, rank=pool.rank, world=pool.world, pooled=pooled, serial=serial, batch=batch, batch_warm=batch_warm,
             alone=alone, alone_warm=alone_warm, one_by_one=one_by_one, mu_pool=g1._mu, mu_alone=g2._mu,
             chain_pool=chains[0][0], chain_alone=chains[1][0], lp_pool=chains[0][1], lp_alone=chains[1][1],
             odd=np.array(odd))
    pool.close()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
