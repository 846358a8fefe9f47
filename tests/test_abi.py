"""The C-ABI library: builds for gfx950, loads, and exports exactly what
include/gprn_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

from gpyrn_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'gprn_hip.h')


def _declared():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gprn_[A-Za-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_hip.SIGNATURES)


def test_library_exports_every_symbol():
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _hip.load_library()
    for name in _declared():
        assert hasattr(lib, name), name


def test_kernel_ids_match_header():
    from gpyrn_amd import covfunc
    text = open(HEADER).read()
    ids = {m.group(1): int(m.group(2))
           for m in re.finditer(r'GPRN_K_([A-Z0-9]+)\s*=\s*(\d+)', text)}
    ids.pop('COUNT')
    assert ids == covfunc.KID
    assert (covfunc.OP_PUSH, covfunc.OP_ADD, covfunc.OP_MUL) == (0, 1, 2)


def test_no_gpu_is_an_error_not_a_fallback():
    if _hip.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_hip.BackendUnavailable):
        _hip.Context(0)
    import numpy as np
    import gpyrn_amd as gpyrn
    t, y, e = np.random.RandomState(0).rand(3, 12)
    g = gpyrn.inference(1, t, y, e)
    g.set_components(gpyrn.SquaredExponential(1, 1), gpyrn.SquaredExponential(1, 1),
                     gpyrn.Constant(0), 0.1)
    with pytest.raises(_hip.BackendUnavailable):
        _ = g.ELBO


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gpyrn_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'cpu_ref' not in src and 'from oracle' not in src and \
                    'import oracle' not in src, f
