"""Loaders for the data tables gpyrn ships (gpyrn/datasets/).

The reference carries one table, ``Solar_observations.txt``: 497 HARPS-N-style
solar observations, one header line and 13 tab-separated columns
(BJD, RV, RVerr, RHK, RHKerr, S, Serr, BIS, BISerr, FWHM, FWHMerr, Constrast,
Contrasterr -- the header's own spelling).  The table itself is not copied into
this package: ``load_solar`` reads it from, in this order, an explicit path,
``$GPYRN_DATASETS``, this directory, or an installed ``gpyrn`` package.
"""
import os

import numpy as np

SOLAR_FILE = 'Solar_observations.txt'
SOLAR_COLUMNS = ('BJD', 'RV', 'RVerr', 'RHK', 'RHKerr', 'S', 'Serr', 'BIS', 'BISerr',
                 'FWHM', 'FWHMerr', 'Constrast', 'Contrasterr')


def _find(name):
    here = os.path.dirname(os.path.abspath(__file__))
    places = [os.environ.get('GPYRN_DATASETS'), here]
    try:
        import importlib.util
        spec = importlib.util.find_spec('gpyrn')
        if spec is not None and spec.submodule_search_locations:
            places.append(os.path.join(list(spec.submodule_search_locations)[0], 'datasets'))
    except (ImportError, ValueError):
        pass
    for d in places:
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    raise FileNotFoundError(
        f'{name} not found; pass its path, or point GPYRN_DATASETS at the directory that holds it '
        f'(the reference keeps it in gpyrn/datasets/)')


def load_table(path):
    """A whitespace/tab-separated table with one header line -> dict column name -> float array."""
    with open(path) as f:
        header = f.readline().split()
    data = np.loadtxt(path, skiprows=1, ndmin=2)
    if data.shape[1] != len(header):
        raise ValueError(f'{path}: {len(header)} column names but {data.shape[1]} columns')
    return {name: np.ascontiguousarray(data[:, i]) for i, name in enumerate(header)}


def load_solar(path=None):
    """The solar observations as a dict of 13 float arrays (497 rows), keyed by the header's names."""
    table = load_table(path or _find(SOLAR_FILE))
    missing = [c for c in SOLAR_COLUMNS if c not in table]
    if missing:
        raise ValueError(f'not the solar table: columns {missing} are missing')
    return table


def inference_args(table, outputs=('RV', 'BIS', 'FWHM'), time='BJD'):
    """``(time, y1, y1err, y2, y2err, ...)`` for ``gpyrn_amd.inference(q, *args)`` from a loaded table:
    each output column with its ``<name>err`` companion (``Constrast`` pairs with ``Contrasterr``)."""
    args = [table[time]]
    for name in outputs:
        err = 'Contrasterr' if name == 'Constrast' else name + 'err'
        args += [table[name], table[err]]
    return tuple(args)
