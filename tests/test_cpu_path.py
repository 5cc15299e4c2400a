"""device='cpu' wrapper against golden outputs of the reference's own CPU path (bit-for-bit: same scipy call)."""
import numpy as np
import pytest

import voltools_amd as vt

INTERPS = ['linear', 'bspline', 'bspline_simple', 'filt_bspline', 'filt_bspline_simple']


@pytest.mark.parametrize('case', ['rot_inplane', 'rot_general', 'rot_scale_shift', 'shear'])
def test_affine_cpu_matches_reference(case, golden_volumes, golden_volume):
    m = golden_volumes[f'{case}/matrix']
    for interp in INTERPS:
        got = vt.affine(golden_volume, m, interpolation=interp, device='cpu')
        assert np.array_equal(got, golden_volumes[f'{case}/{interp}']), (case, interp)


def test_front_ends_cpu(golden_volumes, golden_volume):
    v = golden_volume
    assert np.array_equal(vt.transform(v, rotation=(0, 30, 0), scale=1.2, interpolation='filt_bspline', device='cpu'),
                          golden_volumes['frontend/transform'])
    assert np.array_equal(vt.rotate(v, (15, 25, 35), rotation_order='szyx', interpolation='linear', device='cpu'),
                          golden_volumes['frontend/rotate'])
    assert np.array_equal(vt.translate(v, (2, -1, 3), interpolation='linear', device='cpu'), golden_volumes['frontend/translate'])
    assert np.array_equal(vt.scale(v, 1.5, interpolation='bspline', device='cpu'), golden_volumes['frontend/scale'])
    assert np.array_equal(vt.shear(v, 0.1, interpolation='linear', device='cpu'), golden_volumes['frontend/shear'])
    sv = vt.StaticVolume(v, interpolation='filt_bspline', device='cpu')
    assert np.array_equal(sv.transform(rotation=(0, 60, 0), translation=(1, 2, 3)), golden_volumes['static/transform'])


def test_output_argument_and_reshape_cpu(golden_volumes, golden_volume):
    buf = np.full(golden_volume.shape, 7.0, dtype=np.float32)
    res = vt.affine(golden_volume, golden_volumes['rot_general/matrix'], interpolation='linear', output=buf, device='cpu')
    assert res is buf and np.array_equal(buf, golden_volumes['frontend/output_arg'])
    m = golden_volumes['reshape/matrix']
    pb, pa, nd = vt.utils.compute_post_transform_dimensions(golden_volume.shape, m)
    assert np.array_equal(pb, golden_volumes['reshape/pad_before'])
    assert np.array_equal(pa, golden_volumes['reshape/pad_after'])
    assert np.array_equal(nd, golden_volumes['reshape/new_dims'])
    got = vt.affine(golden_volume, m, interpolation='linear', reshape=True, device='cpu')
    assert np.array_equal(got, golden_volumes['reshape/linear'])


def test_api_surface_and_errors():
    assert vt.AVAILABLE_INTERPOLATIONS == INTERPS
    assert vt.AVAILABLE_DEVICES[0] == 'cpu'
    v = np.zeros((4, 5, 6), np.float32)
    with pytest.raises(ValueError):
        vt.affine(v, np.eye(4, dtype=np.float32), device='tpu')
    with pytest.raises(ValueError):
        vt.StaticVolume(v[0], device='cpu')
    with pytest.raises(ValueError):
        vt.StaticVolume(v, device='gpu:99')
    # unknown interpolation names are silently cubic without prefilter on the CPU path
    assert vt.affine(v, np.eye(4, dtype=np.float32), interpolation='whatever', device='cpu').shape == v.shape
    sv = vt.StaticVolume(v, interpolation='linear', device='cpu')
    assert sv.shape == (4, 5, 6) and sv.device == 'cpu' and sv.interpolation == 'linear'
    for name in ('transform', 'affine', 'rotate', 'scale', 'shear', 'translate', 'StaticVolume', 'utils'):
        assert hasattr(vt, name)
    for name in ('transform_matrix', 'rotation_matrix', 'scale_matrix', 'shear_matrix', 'translation_matrix',
                 'get_available_devices', 'switch_to_device', 'compute_post_transform_dimensions',
                 'compute_prefilter_workgroup_dims', 'compute_elementwise_launch_dims',
                 'AVAILABLE_ROTATIONS', 'AVAILABLE_UNITS'):
        assert hasattr(vt.utils, name)


def test_informational_launch_geometry_helpers():
    """The reference exports its CUDA launch-geometry helpers from voltools.utils (utils/__init__.py:4-5, general.py:9-58); the names exist
    here with the same results (the prefilter one against outputs of the reference itself, tests/golden/launch_dims.npz; the elementwise
    one against the reference's formula evaluated by hand for wavefront size 64 and 256 compute units)."""
    import os
    from voltools_amd.utils import compute_elementwise_launch_dims, compute_prefilter_workgroup_dims
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'launch_dims.npz'))
    for shape, grids, blocks in zip(g['shapes'], g['grids'], g['blocks']):
        got_g, got_b = compute_prefilter_workgroup_dims(tuple(int(n) for n in shape))
        assert np.array_equal(np.array(got_g), grids) and np.array_equal(np.array(got_b), blocks), shape
    try:
        from voltools_amd import _native
        cus = _native.device_props(0)[0] if _native.device_count() > 0 else 256
    except OSError:
        cus = 256
    cap = 32 * cus
    assert compute_elementwise_launch_dims((1, 1, 1)) == ((1, 1, 1), (64, 1, 1))                    # fewer voxels than a wavefront
    assert compute_elementwise_launch_dims((3, 4, 11)) == ((3, 1, 1), (64, 1, 1))                   # one wavefront per 64 voxels
    n = cap * 64 + 1
    assert compute_elementwise_launch_dims((1, 1, n)) == ((cap, 1, 1), (128, 1, 1))                 # all blocks, two wavefronts each
    assert compute_elementwise_launch_dims((512, 512, 512)) == ((cap, 1, 1), (128, 1, 1))           # the grid-stride regime
