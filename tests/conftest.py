import os
import sys

# before anything imports torch (the HIP runtime reads it when it is loaded): ROCm 7.0's graph AQL-packet capture faults when kernels
# of a captured graph are also launched eagerly between replays (bench.py, tools/graph_dist_probe.py)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')


@pytest.fixture
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz')))
    return load


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
