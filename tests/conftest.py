import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# a fatal glibc diagnostic (heap consistency check, abort message) goes to
# /dev/tty unless this is set -- an abort in the native library must leave its
# message in the captured stderr
os.environ.setdefault('LIBC_FATAL_STDERR_', '1')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def toy_prob():
    import scenarios
    return scenarios.toy_problem()
