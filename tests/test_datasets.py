"""The real-data path: loader for the reference's solar table, pinned by the
per-column sums and end rows recorded from the file itself (tests/golden/solar.json)."""
import json
import os

import numpy as np
import pytest

from gpyrn_amd import datasets

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'solar.json')
REF_FILE = '/root/reference/gpyrn/datasets/Solar_observations.txt'


def _write_table(path, header, rows):
    with open(path, 'w') as f:
        f.write('\t'.join(header) + '\n')
        for r in rows:
            f.write('\t'.join(repr(float(v)) for v in r) + '\n')


def test_loader_on_a_table_of_the_same_format(tmp_path):
    rng = np.random.RandomState(3)
    rows = rng.uniform(-5, 5, size=(17, 13))
    p = tmp_path / datasets.SOLAR_FILE
    _write_table(p, datasets.SOLAR_COLUMNS, rows)
    t = datasets.load_solar(str(p))
    assert list(t.keys()) == list(datasets.SOLAR_COLUMNS)
    for i, name in enumerate(datasets.SOLAR_COLUMNS):
        np.testing.assert_array_equal(t[name], rows[:, i])
    args = datasets.inference_args(t, outputs=('RV', 'Constrast'))
    assert len(args) == 5 and args[0] is t['BJD'] and args[4] is t['Contrasterr']
    # found through the environment as well
    os.environ['GPYRN_DATASETS'] = str(tmp_path)
    try:
        assert datasets.load_solar()['RV'].shape == (17,)
    finally:
        del os.environ['GPYRN_DATASETS']


def test_wrong_table_is_refused(tmp_path):
    p = tmp_path / 'other.txt'
    _write_table(p, ['a', 'b'], [[1, 2], [3, 4]])
    with pytest.raises(ValueError):
        datasets.load_solar(str(p))


@pytest.mark.skipif(not os.path.exists(REF_FILE), reason='the reference tree is not on this box')
def test_solar_table_of_the_reference():
    with open(GOLDEN) as f:
        want = json.load(f)
    t = datasets.load_solar(REF_FILE)
    assert list(t.keys()) == want['header'] and os.path.getsize(REF_FILE) == want['bytes']
    cols = np.column_stack([t[k] for k in want['header']])
    assert list(cols.shape) == want['shape'] == [497, 13]
    np.testing.assert_allclose(cols.sum(axis=0), want['colsum'], rtol=1e-13)
    np.testing.assert_array_equal(cols[0], want['first'])
    np.testing.assert_array_equal(cols[-1], want['last'])
    args = datasets.inference_args(t)
    assert len(args) == 7 and all(a.shape == (497,) for a in args)
