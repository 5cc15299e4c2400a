import os
import sys

import numpy as np
import pytest

os.environ.setdefault('LIBC_FATAL_STDERR_', '1')      # glibc's heap / stack check messages to stderr, not /dev/tty
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _ensure_built():
    lib = os.path.join(ROOT, 'voltools_amd', 'lib', 'libvoltools_hip.so')
    orc = os.path.join(ROOT, 'oracle', '_build', 'libvt_oracle.so')
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


_ensure_built()


@pytest.fixture(scope='session')
def golden_volumes():
    return np.load(os.path.join(GOLDEN, 'volumes.npz'))


@pytest.fixture(scope='session')
def golden_m12():
    """48x52x56 fixture: reference CPU path outputs for filt_bspline, pinned at margin 12 (make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, 'volumes_m12.npz'))
    shape = tuple(int(s) for s in g['shape'])
    vol = np.random.RandomState(int(g['seed'])).random_sample(shape).astype(np.float32)
    return g, vol


@pytest.fixture(scope='session')
def golden_matrices():
    return np.load(os.path.join(GOLDEN, 'matrices.npz'))


@pytest.fixture(scope='session')
def golden_volume(golden_volumes):
    shape = tuple(int(s) for s in golden_volumes['shape'])
    return np.random.RandomState(int(golden_volumes['seed'])).random_sample(shape).astype(np.float32)


def source_coords(m, shape):
    """float64 source coordinate of every output voxel, shape (*shape, 3)."""
    g = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing='ij'), -1)
    m = np.asarray(m, dtype=np.float64)
    return g @ m[:3, :3].T + m[:3, 3]


def interior_mask(m, out_shape, src_shape, margin):
    """Output voxels whose source coordinate lies in [margin, dim-1-margin] on every axis -- the region
    where the reference's GPU contract (zero border) and CPU contract (scipy mode='constant') agree
    (SURVEY.md section 8c)."""
    s = source_coords(m, out_shape)
    hi = np.asarray(src_shape, dtype=np.float64) - 1 - margin
    return np.all((s >= margin) & (s <= hi), axis=-1)
