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
    # one object, warm-started, through the public entry point
    batch = fresh().nELBO_batch(sets, pool=pool)
    # tuples, -inf and nan travel unchanged
    odd = pool.map(lambda i: (float(i), -np.inf if i == 1 else (np.nan if i == 2 else 0.5 * i)), range(7))
    np.savez(out, rank=pool.rank, world=pool.world, pooled=pooled, serial=serial, batch=batch,
             odd=np.array(odd))
    pool.close()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
