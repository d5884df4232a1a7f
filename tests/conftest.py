import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def golden_grad_slices(g):
    """(parameter name, row step, reference rows) of the sliced gradients a training golden may carry (`gslice_*`: every n-th row of a
    large recurrent matrix's gradient, tools/gen_golden.py MC4_GSLICES)."""
    if 'gslice_keys' not in g:
        return []
    return [(str(k), int(st), g[f'gslice_{i}']) for i, (k, st) in enumerate(zip(g['gslice_keys'], g['gslice_step']))]
