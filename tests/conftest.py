import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def group():
    from roreg_amd.group import tables
    return tables()


# ROREG_POISON_EMPTY=1: every device tensor that torch.empty / empty_like hands out is filled with NaN (floats) or 0x7f bytes (integers)
# first, so that a kernel relying on uninitialised workspace or output memory shows up as a failing test instead of an intermittent one
# (the caching allocator otherwise returns whatever an earlier tensor left behind).  Diagnostic switch, off by default.
if os.environ.get('ROREG_POISON_EMPTY'):
    import torch as _torch
    _empty, _empty_like = _torch.empty, _torch.empty_like

    def _poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float('nan'))
            else:
                t.view(_torch.uint8).fill_(0x7f) if t.is_contiguous() else None
        return t

    _torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    _torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
