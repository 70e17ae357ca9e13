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


def canon_knn(d, i):
    """Rows of a k-nearest list re-ordered by (distance, index): torch.topk (the reference's find_knn_gpu, utils/knn_search.py:84) returns
    exactly tied entries in an unspecified order, ours in index order -- the lists are compared in this canonical order."""
    d = np.asarray(d); i = np.asarray(i)
    order = np.lexsort((i, d), axis=-1)
    return np.take_along_axis(d, order, -1), np.take_along_axis(i, order, -1)


def same_knn_up_to_duplicates(i, ri, targets):
    """k-nearest index lists (canonical order) equal except where the two name byte-identical target rows (an exact tie at the list's
    k-th place: which duplicate makes the list is torch.topk's unspecified choice in the reference)."""
    i = np.asarray(i); ri = np.asarray(ri)
    bad = i != ri
    return bool(np.array_equal(targets[i[bad]], targets[ri[bad]])), int(bad.sum())
