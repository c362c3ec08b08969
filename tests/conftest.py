import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (run on the MI355X box)')


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run on a machine without a GPU must fail loudly, not skip:
    # the product has no CPU path.  (`-m "not gpu"` deselects them instead.)
    pass


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def hbb_to_rbox(b, rot=False):
    """xyxy(+score) -> theta-OBB with the +1 pixel convention.
    rot=True describes the same rectangle with theta = -pi/2 and w,h swapped."""
    b = np.asarray(b, dtype=np.float64)
    w = b[:, 2] - b[:, 0] + 1
    h = b[:, 3] - b[:, 1] + 1
    cx = (b[:, 0] + b[:, 2]) / 2
    cy = (b[:, 1] + b[:, 3]) / 2
    if rot:
        cols = [cx, cy, h, w, np.full(len(b), -np.pi / 2)]
    else:
        cols = [cx, cy, w, h, np.zeros(len(b))]
    if b.shape[1] > 4:
        cols.append(b[:, 4])
    return np.stack(cols, 1).astype(np.float32)


@pytest.fixture
def nms_route():
    """`nms_route('fused' | 'direct' | 'sorted' | 'listed' | 'grid' | 'auto')` forces the NMS code path through
    the library's tuning hooks (include/obbops.h, obb_set_tuning); defaults are restored afterwards.
    'sorted' = spatial sort with the candidates clipped in line by the scan, 'listed' = spatial sort
    with the candidate lists and the separate clip launch (the default above 65 536 boxes)."""
    from aidet_amd import _lib

    def force(route):
        fused = {'fused': 1024, 'direct': 0, 'sorted': 0, 'listed': 0, 'grid': 0, 'auto': -1}[route]
        direct = {'fused': -1, 'direct': 65536, 'sorted': 0, 'listed': 0, 'grid': 0, 'auto': -1}[route]
        inline = {'fused': -1, 'direct': -1, 'sorted': 1 << 40, 'listed': 0, 'grid': 0, 'auto': -1}[route]
        # 'grid': the uniform-grid route wherever it applies (ungrouped float calls of 256 .. 2^18 - 1 boxes through the
        # synchronous entry points; a scene it gives up on is repeated on the sorted route); off for 'sorted' / 'listed'
        grid = {'fused': -1, 'direct': -1, 'sorted': 0, 'listed': 0, 'grid': 1, 'auto': -1}[route]
        _lib.set_tuning(_lib.TUNE_NMS_FUSED_MAX, fused)
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, direct)
        _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, inline)
        _lib.set_tuning(_lib.TUNE_NMS_GRID, grid)
    yield force
    force('auto')
