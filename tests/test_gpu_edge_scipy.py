"""-m gpu: the CPU path's boundary contract on the GPU (edge='scipy' / VT_EDGE_SCIPY).

The reference's only statement of cpu/gpu equivalence is a pair of plots (`/root/reference/tests/test_devices.py:43-77`).  With
edge='scipy' the HIP path reproduces `scipy.ndimage.affine_transform(mode='constant', cval=0)` -- the reference's CPU path,
`transforms.py:147-152` -- on the WHOLE volume: hard cut-off outside [0, dim-1], mirrored taps, mirror-boundary prefilter.
Checked against the reference's own outputs (golden fixtures) and against this package's device='cpu' path (the same scipy
call, bit-identical to the reference on the fixtures) on ragged shapes, with every kernel family forced.
Tolerances: float32 interpolation weights against scipy's float64 ones, and for filt_* the float32 prefilter recursion."""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native

pytestmark = pytest.mark.gpu
TOL = {'linear': 1e-6, 'bspline': 2e-6, 'bspline_simple': 2e-6, 'filt_bspline': 1e-5, 'filt_bspline_simple': 1e-5}
FLAG_SETS = (0, _native.FORCE_TILED, _native.FORCE_TILED | _native.NO_QUAD, _native.FORCE_TILED | _native.NO_ZPAIR, _native.FORCE_TILED | _native.NO_MARCH,
             _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED, _native.FORCE_TILED | _native.NO_ZSEP | _native.NO_PACKED,
             _native.FORCE_DIRECT)


def centre(shape):
    return np.divide(np.subtract(shape, 1), 2, dtype=np.float32)


@pytest.mark.parametrize('interp', list(TOL))
def test_whole_volume_equals_reference_cpu_path_golden(interp, golden_volumes, golden_volume):
    for case in ('rot_inplane', 'rot_general', 'rot_scale_shift', 'shear'):
        m = golden_volumes[f'{case}/matrix']
        ref = golden_volumes[f'{case}/{interp}']
        sv = vt.StaticVolume(golden_volume, interpolation=interp, device='gpu:0', edge='scipy')
        for flags in FLAG_SETS:
            got = sv.affine(m, _flags=flags)
            err = float(np.abs(got - ref).max())
            assert err <= TOL[interp], (interp, case, flags, sv.info().last_kernel, err)
        sv.close()
        # the functional front end (one-shot path)
        got = vt.affine(golden_volume, m, interpolation=interp, device='gpu', edge='scipy')
        assert np.abs(got - ref).max() <= TOL[interp], (interp, case, 'one-shot')


@pytest.mark.parametrize('shape', [(33, 47, 50), (70, 66, 72), (5, 9, 130), (1, 20, 24), (140, 150, 130)])
@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline'])
def test_whole_volume_equals_cpu_device(interp, shape):
    vol = np.random.RandomState(5).random_sample(shape).astype(np.float32)
    c = centre(shape)
    mats = [vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(0.5, -1.25, 2.0), center=c),
            vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', scale=(0.9, 1.1, 0.95), center=c),
            vt.utils.translation_matrix((3, -2, 5)), np.eye(4, dtype=np.float32),
            vt.utils.transform_matrix(rotation=(0, 0, 120), rotation_order='sxyz', center=c)]
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0', edge='scipy')
    for i, m in enumerate(mats):
        want = vt.affine(vol, m, interpolation=interp, device='cpu')
        for flags in (0, _native.FORCE_TILED, _native.FORCE_DIRECT):
            got = sv.affine(m, _flags=flags)
            err = float(np.abs(got - want).max())
            assert err <= TOL[interp], (interp, shape, i, flags, sv.info().last_kernel, err)
    # the projection of such a handle: transform, then sum
    want = vt.affine(vol, mats[0], interpolation=interp, device='cpu').astype(np.float64).sum(axis=0)
    assert np.abs(sv.projection(mats[0]) - want).max() <= TOL[interp] * shape[0]
    sv.close()


def test_texture_contract_is_unchanged_and_slabs_are_refused():
    import ctypes
    lib = _native.load()
    vol = np.random.RandomState(1).random_sample((20, 24, 28)).astype(np.float32)
    h = ctypes.c_void_p()
    rc = lib.vt_volume_create_slab(0, 20, 24, 28, 0, vol.ctypes.data, _native.EDGE_SCIPY, 4, 40, 4, 20, ctypes.byref(h))
    assert rc != 0                       # slab windows keep the texture contract
    with pytest.raises(ValueError):
        vt.StaticVolume(vol, device='gpu', edge='mirror')
