"""Rebuild model problems from the committed golden fixtures (data only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(tag):
    with open(os.path.join(GOLDEN, tag + '.json')) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, tag + '.npz'))
    return meta, data


def available(tag):
    return os.path.exists(os.path.join(GOLDEN, tag + '.npz'))


def components(meta, covfunc, meanfunc):
    mk = lambda mod, item: None if item is None else getattr(mod, item[0])(*item[1])
    nodes = [mk(covfunc, n) for n in meta['nodes']]
    weights = [mk(covfunc, w) for w in meta['weights']]
    means = [mk(meanfunc, m) for m in meta['means']]
    return nodes, weights, means, list(meta['jitters'])


def data_args(data):
    args = []
    for y, e in zip(data['y'], data['yerr']):
        args += [np.array(y), np.array(e)]
    return args
