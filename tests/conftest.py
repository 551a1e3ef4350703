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
    config.addinivalue_line('markers', 'slow: CPU test that takes more than a few seconds')


def golden(name):
    """Load a fixture recorded from the reference by tests/golden/make_golden.py."""
    path = os.path.join(GOLDEN, name + '.npz')
    if not os.path.exists(path):
        pytest.skip(f'fixture {name}.npz missing')
    return dict(np.load(path))


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
