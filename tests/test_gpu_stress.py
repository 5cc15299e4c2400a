"""-m gpu: handle life-cycle stress.

10^4 cycles of create -> transform into a pooled host buffer -> destroy over interleaved sizes, with the library's device-buffer
cache (`cached_malloc` / `cached_free`, eviction, `vt_device_trim`), the stream / event free lists, the lazily built resident
copies and the host result pool (register / unregister of pinned buffers) all churning; results are compared bit for bit with
the first result of the same (shape, interpolation), which is itself checked against the oracle.  Sizes straddle the pool
threshold (1 MiB results) and the in-place pinning threshold (8 MiB uploads).  The round-1 abort (a GPU memory access fault
on a HOST address: overlapping page-granular registrations) came out of exactly this churn."""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_create_transform_destroy_10k_cycles():
    rs = np.random.RandomState(2024)
    shapes = [(64, 64, 64), (66, 70, 72), (40, 96, 80), (30, 30, 30), (96, 100, 104), (128, 128, 130), (64, 66, 64)]
    weights = np.array([6, 6, 6, 4, 2, 1, 6], dtype=np.float64)
    weights /= weights.sum()
    interps = ['linear', 'bspline', 'filt_bspline']
    vols = {s: rs.random_sample(s).astype(np.float32) for s in shapes}
    mats = {s: vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(0.5, -1.25, 2.0),
                                         center=np.divide(np.subtract(s, 1), 2, dtype=np.float32)) for s in shapes}
    gen = {s: vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz',
                                        center=np.divide(np.subtract(s, 1), 2, dtype=np.float32)) for s in shapes}
    expected = {}
    held = []
    cycles = 10000
    for i in range(cycles):
        s = shapes[rs.choice(len(shapes), p=weights)]
        interp = interps[i % 3]
        m = mats[s] if (i // 3) % 2 == 0 else gen[s]
        key = (s, interp, (i // 3) % 2)
        sv = vt.StaticVolume(vols[s], interpolation=interp, device='gpu:0')
        got = sv.affine(m)
        sv.close()
        if key not in expected:
            tol = 1e-5 if interp.startswith('filt') else 2e-6
            assert np.abs(got - oracle.affine(vols[s], m, interp)).max() <= tol, key
            expected[key] = got.copy()
        else:
            assert np.array_equal(got, expected[key]), (i, key)
        if i % 5 == 0:                                  # callers that keep results: the pool must not hand those buffers out again
            held.append((key, got))
            if len(held) > 6:
                k0, old = held.pop(rs.randint(len(held)))
                assert np.array_equal(old, expected[k0]), (i, k0, 'a held result was overwritten')
        if i % 997 == 996:
            _native.free_cached_memory(0)
    for k0, old in held:
        assert np.array_equal(old, expected[k0])
    _native.free_cached_memory(0)
