"""One rank of a sharded run on a golden case (started by test_parity_gpu.py).

usage: python -m tests._shard_worker <tag> <out.npz> <rendezvous tag>
with RANK / WORLD_SIZE / LOCAL_RANK in the environment.
"""
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
    g.set_components(nodes, weights, means, jit)
    res = {}
    # forced sweeps from the reference's own initial state
    ctx = g._setup_device(g.nodes, g.weights, g.means, g.jitters)
    ctx.set_muvar(d['mu_init'], d['var_init'])
    elbo, parts, info = ctx.sweep(meta['nsweeps'], commit=True)
    mu, var = ctx.get_muvar()
    res.update(sw_elbo=elbo, sw_parts=parts, sw_info=info, sw_mu=mu, sw_var=var,
               logdet_K=ctx.get_logdet_K())
    if 'calc_elbo' in d:
        E, mu, var, it = g.ELBOcalc()
        res.update(calc_elbo=E, calc_mu=mu, calc_var=var, calc_iter=it,
                   calc_history=np.array(g._elbo_history))
    np.savez(out, rank=comm.rank, world=comm.world, **res)
    comm.cleanup()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
