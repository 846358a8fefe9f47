import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_sessionfinish(session, exitstatus):
    """The norm-wise deviations of the posterior state that tests/_cases.assert_state saw, fixture by fixture."""
    from tests import _cases
    if not _cases.ACHIEVED:
        return
    out_dir = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        gpu = any('gpu' in (getattr(item, 'keywords', {}) or {}) for item in getattr(session, 'items', []))
        name = 'parity_achieved.txt' if gpu else 'parity_achieved_cpu.txt'
        with open(os.path.join(out_dir, name), 'w') as f:
            f.write('# max |x - x_ref| / max |x_ref| per latent GP (worst row), bound %.0e: posterior means | variances\n'
                    % _cases.STATE_TOL)
            for what, e_mu, e_var in _cases.ACHIEVED:
                f.write('%-60s %.3e  %s\n' % (what, e_mu, '-' if e_var is None else '%.3e' % e_var))
    except OSError:
        pass
