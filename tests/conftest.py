import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def has_f16():
    """The library carries the optional half-operand kernel twins (built with AMTX_BUILD_F16=1; off by default since round 6)."""
    try:
        from amt_tools_amd import _lib
        return bool(_lib.lib().amtx_has_f16())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    import re
    f16 = None
    for item in items:
        if re.search(r'(^|[\[\-_])f16($|[\]\-_])', item.name):
            if f16 is None:
                f16 = has_f16()
            if not f16:
                item.add_marker(pytest.mark.skip(reason="precision 'f16' is an optional build (AMTX_BUILD_F16=1 python -m amt_tools_amd.build)"))


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
