"""-m gpu: parity on data that is NOT uniform [0, 1).

The reference's texture is plain float32 of any range (`/root/reference/voltools/transforms.py:184`: `cp.asarray(volume)` as it
comes); every other GPU test of this suite draws `random_sample`.  Here: signed normal data at three magnitudes (tolerance relative to
max|volume|: the arithmetic is linear in the data), a constant volume (trilinear: exact wherever all taps are inside), a volume of
subnormal floats (gfx950 keeps float32 subnormals in v_fma / v_pk_fma: the kernels must not flush what the oracle keeps), and
NaN / Inf placement: for every kernel family WHICH output voxels turn non-finite is pinned against the oracle's set -- every output whose
stencil contains the bad voxel, also through a weight of exactly zero (0 * NaN = NaN, `helper_interpolation.h:3-6` has no special
case) -- with the one documented exception of DESIGN section 2: the integer-axis-0-offset instantiations (`KIND 3` trilinear, `KIND 4` cubic)
never read the tap plane whose weight is exactly 0.
"""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

pytestmark = pytest.mark.gpu

TOL = {'linear': 1e-6, 'bspline': 1e-6, 'bspline_simple': 1e-6, 'filt_bspline': 3e-6, 'filt_bspline_simple': 3e-6}
SHAPE = (70, 66, 72)
FT = _native.FORCE_TILED
# (flags, kernels that may serve an axis-0-separable / a general matrix under them)
FAMILIES = ((0, None), (FT, None), (FT | _native.NO_RSWAP, None), (FT | _native.NO_QUAD, None), (FT | _native.NO_QUAD | _native.NO_BLOCK, None),
            (FT | _native.NO_ZSEP | _native.FORCE_PACKED, None), (FT | _native.NO_ZSEP | _native.NO_PACKED, None), (FT | _native.NO_ZSEP, None),
            (_native.FORCE_DIRECT, None))


def centre(shape):
    return np.divide(np.subtract(shape, 1), 2, dtype=np.float32)


def matrices(shape):
    c = centre(shape)
    return {
        'sweep33': vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='rzxz', center=c),
        'sweep100_shift': vt.utils.transform_matrix(rotation=(0, 100, 0), rotation_order='rzxz', translation=(1.25, -0.5, 2.0), center=c),
        'general': vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=c),
        'affine': vt.utils.transform_matrix(rotation=(10, 20, 30), scale=(1.1, 0.9, 1.25), translation=(1.5, -2.0, 0.75), center=c),
        'axis1': vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='sxyz', center=c),
        'axis2': vt.utils.transform_matrix(rotation=(0, 0, -120), rotation_order='sxyz', translation=(1.5, -2.25, 0.75), center=c),
        'shift_int': vt.utils.translation_matrix((3, -2, 5)),
        'axis2_w4': vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', translation=(0, 0, 4), center=c),
    }


@pytest.mark.parametrize('interp', list(TOL))
@pytest.mark.parametrize('scale', [1.0, 1e3, 1e-3])
def test_signed_normal_data_relative_tolerance(interp, scale):
    rs = np.random.RandomState(7)
    vol = (rs.standard_normal(SHAPE) * scale).astype(np.float32)
    tol = TOL[interp] * float(np.abs(vol).max())
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for name, m in matrices(SHAPE).items():
        want = oracle.affine(vol, m, interp)
        for flags, _ in FAMILIES:
            got = sv.affine(m, _flags=flags)
            err = float(np.abs(got - want).max())
            assert err <= tol, (interp, scale, name, flags, sv.info().last_kernel, err, tol)
    sv.close()


@pytest.mark.parametrize('value', [1.0, -3.75, 1e6 + 0.5])
def test_constant_volume(value):
    vol = np.full(SHAPE, value, dtype=np.float32)
    v32 = np.float32(value)
    for interp in ('linear', 'bspline', 'filt_bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for name, m in matrices(SHAPE).items():
            want = oracle.affine(vol, m, interp)
            # voxels all of whose taps lie inside the volume (margin 2 for the cubic stencil, 14 more for the prefilter's boundary)
            g = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float64) for s in SHAPE], indexing='ij'), -1)
            s = g @ np.asarray(m, np.float64)[:3, :3].T + np.asarray(m, np.float64)[:3, 3]
            margin = {'linear': 0.0, 'bspline': 1.0, 'filt_bspline': 16.0}[interp]
            inner = np.all((s >= margin) & (s <= np.asarray(SHAPE) - 1.0 - margin), axis=-1)
            for flags, _ in FAMILIES:
                got = sv.affine(m, _flags=flags)
                assert np.abs(got - want).max() <= TOL[interp] * abs(value), (interp, name, flags)
                if interp == 'linear':
                    # fma(f, c - c, c) == c: exact, on every family
                    assert np.array_equal(got[inner], np.full(int(inner.sum()), v32)), (name, flags, sv.info().last_kernel)
                else:
                    assert np.abs(got[inner] - v32).max() <= 4e-6 * abs(value), (interp, name, flags)
        sv.close()


def test_subnormal_volume_is_not_flushed():
    """All samples below FLT_MIN: the oracle (host float32 arithmetic, subnormals kept) and the kernels must agree to a few subnormal
    ulps (1.4e-45 each); a kernel that flushed subnormal inputs or results to zero would return 0 everywhere."""
    rs = np.random.RandomState(11)
    vol = (rs.random_sample(SHAPE) * 1.0e-38).astype(np.float32)           # FLT_MIN = 1.1755e-38
    assert float(vol.max()) < 1.1755e-38 and float(vol.max()) > 0
    ulp = 1.4012984643e-45
    for interp in ('linear', 'bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for name, m in matrices(SHAPE).items():
            want = oracle.affine(vol, m, interp)
            assert float(np.abs(want).max()) > 1e-39
            for flags, _ in FAMILIES:
                got = sv.affine(m, _flags=flags)
                err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
                assert err <= 24 * ulp, (interp, name, flags, sv.info().last_kernel, err / ulp)
        sv.close()


def stencil_hits(m, shape, bad, interp):
    """Output voxels whose interpolation stencil (float64 coordinates, as the kernels and the oracle compute them) contains voxel `bad`
    AND that pass the skirt test."""
    g = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing='ij'), -1)
    m = np.asarray(m, np.float64)
    s = g @ m[:3, :3].T + m[:3, 3]
    inside = np.all((s + 0.5 >= 0) & (s + 0.5 < np.asarray(shape, np.float64)), axis=-1)
    f = np.floor(s).astype(np.int64)
    lo, hi = (0, 1) if interp == 'linear' else (-1, 2)
    hit = inside.copy()
    for a in range(3):
        hit &= (bad[a] >= f[..., a] + lo) & (bad[a] <= f[..., a] + hi)
    return hit


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'bspline_simple'])
@pytest.mark.parametrize('badval', [np.nan, np.inf, -np.inf])
def test_nonfinite_placement(interp, badval):
    rs = np.random.RandomState(3)
    vol = rs.random_sample(SHAPE).astype(np.float32)
    bad = (31, 29, 40)
    vol[bad] = badval
    clean = vol.copy()
    clean[bad] = 0.5
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for name, m in matrices(SHAPE).items():
        want = oracle.affine(vol, m, interp)
        want_bad = ~np.isfinite(want)
        # the oracle's own set is the stencil set: it has no zero-weight shortcut
        assert np.array_equal(want_bad, stencil_hits(m, SHAPE, bad, interp)), name
        ref_clean = oracle.affine(clean, m, interp)
        m64 = np.asarray(m, np.float64)
        for flags, _ in FAMILIES:
            got = sv.affine(m, _flags=flags)
            k = int(sv.info().last_kernel)
            got_bad = ~np.isfinite(got)
            allowed_missing = np.zeros(SHAPE, bool)
            if k in (10, 8):
                # (kind 10, the row kernel: the same along axis 2)
                # KIND 3 / KIND 4 (on the plain copy, or on an axis-exchanged one for rotations about axis 1 / 2): along the marching axis `a`
                # (row a of the matrix is a unit row with an integer offset) trilinear output slice d reads source slice d + off only, the
                # cubic one slices d + off - 1 .. d + off + 1 (through the z-convolved copy); the oracle also multiplies the next slice
                # (d + off + 1, resp. d + off + 2) by exactly 0
                for a in range(3):
                    unit = all(m64[a, c] == (1.0 if c == a else 0.0) for c in range(3))
                    if unit and m64[a, 3] == np.floor(m64[a, 3]):
                        dz = bad[a] - int(m64[a, 3]) - (1 if interp == 'linear' else 2)
                        if 0 <= dz < SHAPE[a]:
                            idx = [slice(None)] * 3
                            idx[a] = dz
                            allowed_missing[tuple(idx)] = want_bad[tuple(idx)]
                        break
            assert not np.any(got_bad & ~want_bad), (interp, name, flags, k, 'non-finite where the oracle is finite')
            missing = want_bad & ~got_bad
            assert not np.any(missing & ~allowed_missing), (interp, name, flags, k, int(missing.sum()))
            if k in (10, 8) and allowed_missing.any():
                assert np.array_equal(missing, allowed_missing), (name, flags)       # pinned: exactly that plane's hits stay finite
                # ... and hold the value the zero weight implies
                sel = allowed_missing
                assert np.abs(got[sel] - ref_clean[sel]).max() <= TOL[interp]
            ok = ~want_bad
            assert np.abs(got[ok] - want[ok]).max() <= TOL[interp], (interp, name, flags, k)
    sv.close()
