import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def input_checksum(cov):
    w = (np.arange(cov.size, dtype=np.float64) % 251.) + 1.
    return float((np.asarray(cov, dtype=np.float64).reshape(-1) * w).sum())


@pytest.fixture(scope='session')
def oracle():
    """The CPU parity oracle (oracle/), compiled on demand.  Test infrastructure only."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope='session')
def device():
    """A degnorm_amd Device on cuda:0; the HIP library must already be built (it travels with the repo)."""
    from degnorm_amd import _lib
    dev = _lib.Device(0)
    yield dev
    dev.close()
