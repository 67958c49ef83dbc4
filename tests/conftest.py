import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip('fixture %s not generated' % name)
    return np.load(path)


@pytest.fixture(scope='session')
def g_small():
    return load_golden('g_small.npz')


@pytest.fixture(scope='session')
def g_mid():
    return load_golden('g_mid.npz')
