"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (unit-range float32 data, stated here and in DESIGN.md):
  * unfiltered interpolations: |hip - oracle| <= 1e-6 everywhere, on every kernel family (same float64 coordinates, same
    float32 weights; only the summation order differs; SURVEY 8c states this figure);
  * filt_*: |hip - oracle| <= 3e-6 everywhere (the coefficients reach +-2 and the three prefilter passes amplify float32
    rounding; the wave-scan / block evaluation re-associates the recursion);
  * against the reference CPU path (scipy, golden fixtures) on the interior mask: 2e-6 / 5e-6.
"""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle
from conftest import interior_mask

pytestmark = pytest.mark.gpu

TOL = {'linear': 1e-6, 'bspline': 1e-6, 'bspline_simple': 1e-6, 'filt_bspline': 3e-6, 'filt_bspline_simple': 3e-6}
ALL_INTERPS = list(TOL)
try:        # the test build (tests/test_gpu_legacy.py loads it through VT_LIB) also carries round 1's kernels 3, 4, 5
    LEGACY = _native.has_legacy_kernels()
except Exception:  # pragma: no cover  (library not built: the gpu tests are skipped or fail on their own)
    LEGACY = False
GENERAL_KERNELS = (2, 6, 9)      # what serves an axis-0-separable matrix that is kept off the plane-quad kernel in the product build


def rand_vol(shape, seed=0):
    return np.random.RandomState(seed).random_sample(shape).astype(np.float32)


def centre(shape):
    return np.divide(np.subtract(shape, 1), 2, dtype=np.float32)


MATRICES = {
    'identity': lambda s: np.eye(4, dtype=np.float32),
    'shift_int': lambda s: vt.utils.translation_matrix((3, -2, 5)),
    'shift_frac': lambda s: vt.utils.translation_matrix((0.5, -1.25, 2.75)),
    'rot_inplane45': lambda s: vt.utils.transform_matrix(rotation=(0, 45, 0), center=centre(s)),
    'rot_inplane100': lambda s: vt.utils.transform_matrix(rotation=(0, 100, 0), translation=(0.5, -1.25, 2.0), center=centre(s)),
    'rot_inplane260': lambda s: vt.utils.transform_matrix(rotation=(0, 260, 0), center=centre(s)),
    'rot_general': lambda s: vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=centre(s)),
    'rot_scale_shift': lambda s: vt.utils.transform_matrix(rotation=(10, 20, 30), scale=(1.1, 0.9, 1.25),
                                                           translation=(1.5, -2.0, 0.75), center=centre(s)),
    'shear': lambda s: vt.utils.transform_matrix(shear=(0.1, -0.05, 0.2), center=centre(s)),
    'magnify3': lambda s: vt.utils.transform_matrix(scale=(3.0, 3.0, 3.0), center=centre(s)),
    'minify': lambda s: vt.utils.transform_matrix(scale=(0.4, 0.5, 0.3), center=centre(s)),
    'minify_big': lambda s: vt.utils.transform_matrix(scale=(0.05, 0.05, 0.05), center=centre(s)),
    'far_outside': lambda s: vt.utils.translation_matrix((1e4, 0, 0)),
    'mirror': lambda s: vt.utils.transform_matrix(scale=(-1.0, 1.0, -1.0), center=centre(s)),
    'rot_axis2': lambda s: vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', center=centre(s)),
    'rot_axis2_shift': lambda s: vt.utils.transform_matrix(rotation=(0, 0, -120), rotation_order='sxyz',
                                                           translation=(1.5, -2.25, 0.75), center=centre(s)),
    'rot_axis1': lambda s: vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='sxyz', center=centre(s)),
    'rot_axis1_shift': lambda s: vt.utils.transform_matrix(rotation=(0, -120, 0), rotation_order='sxyz',
                                                           translation=(1.5, -2.25, 0.75), center=centre(s)),
}


def run_case(vol, m, interp, flags=0, keep=False, out_init=None):
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    if out_init is None:
        got = sv.affine(m, _flags=flags)
    else:
        got = out_init.copy()
        sv.affine(m, output=got, keep_outside=keep, _flags=flags)
    info = sv.info()
    sv.close()
    return got, info


@pytest.mark.parametrize('interp', ALL_INTERPS)
@pytest.mark.parametrize('mname', list(MATRICES))
@pytest.mark.parametrize('shape', [(70, 66, 72), (33, 47, 50)])
def test_tiled_and_direct_match_oracle(interp, mname, shape):
    """Every interpolation x matrix on a width divisible by 4 (16-byte staging) and one that is not."""
    vol = rand_vol(shape, 1)
    m = MATRICES[mname](shape)
    want = oracle.affine(vol, m, interp)
    kernels = set()
    for flags in (_native.FORCE_TILED | _native.FORCE_XSWAP, _native.FORCE_TILED | _native.FORCE_XSWAP | _native.NO_ZPAIR,
                  _native.FORCE_TILED | _native.FORCE_XSWAP | _native.NO_QUAD, _native.FORCE_TILED | _native.NO_QUAD | _native.NO_RSWAP,
                  _native.FORCE_TILED, _native.FORCE_TILED | _native.NO_RSWAP, _native.FORCE_TILED | _native.NO_MARCH,
                  _native.FORCE_TILED | _native.NO_ZSEP, _native.FORCE_TILED | _native.NO_ZSEP | _native.NO_BLOCK,
                  _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED, _native.FORCE_TILED | _native.NO_ZSEP | _native.NO_PACKED,
                  _native.FORCE_DIRECT):
        got, info = run_case(vol, m, interp, flags)
        kernels.add(info.last_kernel)
        err = np.abs(got - want).max()
        assert err <= TOL[interp], f'{interp}/{mname}/{shape} flags={flags} kernel={info.last_kernel} err={err}'
    if mname in ('rot_axis1', 'rot_axis1_shift', 'rot_axis2', 'rot_axis2_shift'):
        # rotations about axis 1 / 2 march along an axis-exchanged resident copy
        assert 8 in kernels and (not LEGACY or ((4 in kernels) if interp == 'linear' else (5 in kernels and 4 in kernels)))
    if mname == 'rot_axis2':
        assert 10 in kernels                         # ... and, with an integer offset along axis 2, the row kernel serves them on the plain copy
    if mname in ('identity', 'shift_int', 'shift_frac', 'rot_inplane45', 'rot_inplane100', 'rot_inplane260'):
        assert 8 in kernels
        if LEGACY:
            assert 3 in kernels and 4 in kernels          # every axis-0-separable kernel was exercised
            assert (5 in kernels) == (interp != 'linear')   # cubic: the plane-pair marching kernel too
    if mname not in ('minify_big', 'far_outside'):
        assert 2 in kernels and 1 in kernels
    if mname in ('rot_general', 'shear', 'rot_scale_shift', 'minify', 'mirror'):
        assert 6 in kernels                          # packed-footprint kernel for invertible general matrices
    if interp != 'linear' and mname in ('rot_general', 'shear', 'rot_scale_shift', 'mirror', 'rot_axis1', 'rot_axis2_shift', 'rot_inplane45', 'shift_frac'):
        assert 9 in kernels                          # lane-block kernel: every matrix whose tile footprint fits its LDS rows


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline_simple'])
@pytest.mark.parametrize('box', ['0', '1'])
def test_marching_staging_modes(interp, box, monkeypatch):
    """Both staging modes of both marching kernels (bounding boxes / packed row spans), whichever the planner prefers."""
    monkeypatch.setenv('VT_MARCH_BOX', box)
    shape = (70, 66, 72)
    vol = rand_vol(shape, 5)
    for mname in ('shift_frac', 'rot_inplane45', 'rot_inplane100', 'rot_axis1_shift', 'rot_axis2_shift'):
        m = MATRICES[mname](shape)
        want = oracle.affine(vol, m, interp)
        for flags in (_native.FORCE_TILED | _native.FORCE_XSWAP | _native.NO_QUAD, _native.FORCE_TILED | _native.FORCE_XSWAP | _native.NO_ZPAIR):
            got, info = run_case(vol, m, interp, flags)
            assert info.last_kernel in ((4, 5) if LEGACY else GENERAL_KERNELS)
            assert np.abs(got - want).max() <= TOL[interp], (interp, box, mname, flags)


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
@pytest.mark.parametrize('knob', [{'VT_DCH': '8'}, {'VT_DCH': '64'}, {'VT_LA': '2'}, {'VT_LA': '3'}, {'VT_BLK_H': '3', 'VT_BLK_W': '2'},
                                  {'VT_BLK_H': '2', 'VT_BLK_W': '5'},
                                  # round 3: lane <-> pixel mapping, row placement in LDS, chunk depth of the one-plane trilinear kernel
                                  {'VT_QUAD_PERM': '0'}, {'VT_QUAD_ROWS': '-1'}, {'VT_QUAD_ROWS': '5'}, {'VT_ZID_DCH': '8'}, {'VT_QUAD_ZID': '0'},
                                  # chunk layers walked from the last to the first (what every other launch of a handle does)
                                  {'VT_QUAD_PINGPONG': '2'}, {'VT_QUAD_PINGPONG': '0'}, {'VT_QUAD_PINGPONG': '2', 'VT_QUAD_GRID2D': '0'},
                                  {'VT_QUAD_PINGPONG': '2', 'VT_BLK_H': '3', 'VT_BLK_W': '2'}])
def test_marching_schedule_does_not_change_results(interp, knob, monkeypatch):
    """Chunk depth (incl. the round-aware default), ring depth and tile order are schedules: the marching kernels must return
    the same bits for every one of them (each voxel is summed in one fixed order; the tile SIZE is not such a knob: pixel
    coordinates are formed relative to the tile origin, a last-bit difference in the weights).  176 x 200 x 232: ragged tiles in
    both in-plane directions, several chunks, partial blocks at both edges of the blocked order."""
    shape = (176, 200, 232)
    vol = rand_vol(shape, 7)
    for mname in ('rot_inplane45', 'shift_frac'):
        m = MATRICES[mname](shape)
        want = oracle.affine(vol, m, interp)
        for flags, kernels in ((0, (8,)), (_native.NO_QUAD, (4, 5))) if LEGACY else ((0, (8,)),):
            ref, info = run_case(vol, m, interp, flags)
            assert info.last_kernel in kernels
            for k, val in knob.items():
                monkeypatch.setenv(k, val)
            got, info2 = run_case(vol, m, interp, flags)
            for k in knob:
                monkeypatch.delenv(k)
            assert info2.last_kernel in kernels
            assert np.array_equal(got, ref), (interp, knob, mname, flags, float(np.abs(got - ref).max()))
            assert np.abs(ref - want).max() <= TOL[interp]


@pytest.mark.parametrize('interp', ['linear', 'bspline_simple', 'filt_bspline'])
def test_alternate_launches_of_a_handle_return_the_same_bits(interp, monkeypatch):
    """Every other plane-quad launch of a handle walks its chunk layers backwards (the memory-side cache still holds the planes the
    previous launch read last): a schedule, so launches 1, 2, 3 of one matrix must agree bit for bit, on 8-plane chunks (22 layers)
    and on the default depth, for integer and fractional axis-0 offsets."""
    shape = (176, 200, 232)
    vol = rand_vol(shape, 11)
    for dch in ('8', None):
        if dch:
            monkeypatch.setenv('VT_DCH', dch)
            monkeypatch.setenv('VT_ZID_DCH', dch)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for mname in ('rot_inplane45', 'shift_frac'):
            m = MATRICES[mname](shape)
            outs = [sv.affine(m) for _ in range(3)]
            assert sv.info().last_kernel == 8
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), (interp, dch, mname)
            assert np.abs(outs[0] - oracle.affine(vol, m, interp)).max() <= TOL[interp]
        sv.close()
        if dch:
            monkeypatch.delenv('VT_DCH')
            monkeypatch.delenv('VT_ZID_DCH')


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline'])
def test_default_dispatch_uses_tiled_kernel_on_large_volumes(interp):
    shape = (96, 100, 104)
    vol = rand_vol(shape, 2)
    m = MATRICES['rot_general'](shape)
    got, info = run_case(vol, m, interp)
    # a volume this small has too few tiles for the persistent lane-block kernel; forced, cubic interpolations take it
    assert info.last_kernel == 2 and info.last_lds_bytes > 0
    assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    got, info = run_case(vol, m, interp, _native.FORCE_TILED)
    assert info.last_kernel == (2 if interp == 'linear' else 9) and info.last_lds_bytes > 0
    assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    got, info = run_case(vol, m, interp, _native.NO_BLOCK)
    # a volume this small has too few tiles to amortise the packed kernel's per-workgroup set-up: bounding boxes
    assert info.last_kernel == 2 and info.last_lds_bytes > 0
    assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    got, info = run_case(vol, m, interp, _native.FORCE_PACKED)
    assert info.last_kernel == 6 and info.last_lds_bytes > 0
    assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    m = MATRICES['rot_inplane45'](shape)
    got, info = run_case(vol, m, interp)
    assert info.last_kernel == 8
    assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    got, info = run_case(vol, m, interp, _native.NO_QUAD)
    assert info.last_kernel in (((4,) if interp == 'linear' else (5,)) if LEGACY else GENERAL_KERNELS)
    assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp]


@pytest.mark.parametrize('tile', ['1', '2', '3', '4', '5'])
@pytest.mark.parametrize('interp', ['linear', 'bspline'])
def test_plane_quad_tile_configurations_on_ragged_shapes(interp, tile, monkeypatch):
    """Every tile configuration of the plane-quad kernel (the planner prefers the 512-thread ones only on 512^3 / 1024^3 launches) on a
    shape no tile divides: integer and fractional axis-0 offsets (the one- and the two-plane trilinear kernels), angles on both sides of the
    in-plane transposition, against the oracle."""
    monkeypatch.setenv('VT_TILE', tile)
    shape = (41, 70, 150)
    vol = rand_vol(shape, 13)
    c = centre(shape)
    for m in (vt.utils.transform_matrix(rotation=(0, 17, 0), translation=(0.0, 1.5, -2.25), center=c),
              vt.utils.transform_matrix(rotation=(0, 123, 0), translation=(3.0, -0.5, 0.75), center=c),
              vt.utils.transform_matrix(rotation=(0, 38, 0), translation=(0.375, 0.0, 0.0), center=c)):
        got, info = run_case(vol, m, interp, _native.FORCE_TILED)
        # (a forced configuration whose ring does not fit -- the four-pixel cubic tiles at steep angles -- is served by another family)
        assert info.last_kernel == 8 or (interp != 'linear' and tile in ('2', '3')), (interp, tile, info.last_kernel)
        assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp], (interp, tile)


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
def test_secondary_copy_failure_replans_on_the_plain_layout(interp, monkeypatch):
    """A plane-quad copy that cannot be allocated (here: VT_TEST_FAIL_COPY) must not fail the call nor leave a zero-filled copy
    behind: the call is planned again without that family and served from the plain layout, every time."""
    if not LEGACY:
        pytest.skip('the failure-injection hook VT_TEST_FAIL_COPY is compiled into the test build only (tests/test_gpu_legacy.py runs this there)')
    shape = (96, 100, 104)
    vol = rand_vol(shape, 9)
    m = MATRICES['rot_inplane45'](shape)
    want = oracle.affine(vol, m, interp)
    monkeypatch.setenv('VT_TEST_FAIL_COPY', '1')
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    monkeypatch.delenv('VT_TEST_FAIL_COPY')
    for _ in range(2):                            # the second call must not find a half-built copy either
        got = sv.affine(m)
        assert sv.info().last_kernel != 8
        assert np.abs(got - want).max() <= TOL[interp]
    sv.close()
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')       # the same call with the copy available
    got = sv.affine(m)
    assert sv.info().last_kernel == 8 and np.abs(got - want).max() <= TOL[interp]
    sv.close()


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 5, 7), (5, 5, 5), (2, 3, 130), (130, 3, 2), (13, 1, 64)])
@pytest.mark.parametrize('interp', ['linear', 'bspline_simple', 'filt_bspline'])
def test_degenerate_and_ragged_shapes(shape, interp):
    vol = rand_vol(shape, 3)
    for mname in ('identity', 'shift_frac', 'rot_general'):
        m = MATRICES[mname](shape)
        want = oracle.affine(vol, m, interp)
        for flags in (_native.FORCE_TILED, _native.FORCE_TILED | _native.NO_QUAD, _native.FORCE_TILED | _native.NO_ZPAIR, _native.FORCE_TILED | _native.NO_MARCH,
                      _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED, _native.FORCE_TILED | _native.NO_ZSEP | _native.NO_PACKED,
                      _native.FORCE_DIRECT):
            got, _ = run_case(vol, m, interp, flags)
            assert np.abs(got - want).max() <= TOL[interp], (shape, interp, mname, flags)


@pytest.mark.parametrize('interp', ALL_INTERPS)
def test_golden_reference_cpu_path_interior(interp, golden_volumes, golden_volume):
    """HIP output vs the reference's own CPU path (scipy), where the two boundary contracts agree."""
    margin = {'linear': 0, 'bspline': 1, 'bspline_simple': 1}.get(interp, 12)
    tol = 2e-6 if not interp.startswith('filt') else 5e-6
    for case in ('rot_inplane', 'rot_general', 'rot_scale_shift', 'shear'):
        m = golden_volumes[f'{case}/matrix']
        ref = golden_volumes[f'{case}/{interp}']
        mask = interior_mask(m, ref.shape, golden_volume.shape, margin)
        if interp.startswith('filt'):
            # a 20x24x28 volume has no voxel 12 away from every face; use the largest margin that keeps some
            margin2 = 8
            mask = interior_mask(m, ref.shape, golden_volume.shape, margin2)
            tol = 2e-5     # |z|^8 = 2.7e-5 of boundary influence remains at margin 8
        assert mask.sum() > 50
        got, _ = run_case(golden_volume, m, interp, _native.FORCE_TILED)
        assert np.abs(got - ref)[mask].max() <= tol, (interp, case)


@pytest.mark.parametrize('interp', ['filt_bspline', 'filt_bspline_simple'])
def test_golden_reference_margin12(interp, golden_m12):
    """The prefiltered interpolations against the reference's own CPU path at the tolerance SURVEY 8c states: 2e-6 where the
    source coordinate is 12 samples away from every face (48x52x56 fixture), on every kernel family."""
    g, vol = golden_m12
    for case in ('rot_inplane', 'rot_general', 'rot_scale_shift'):
        m = g[f'{case}/matrix']
        ref = g[f'{case}/filt_bspline']
        mask = interior_mask(m, ref.shape, vol.shape, 12)
        assert mask.sum() > 10000
        kernels = set()
        for flags in (0, _native.FORCE_TILED, _native.FORCE_TILED | _native.NO_QUAD, _native.FORCE_TILED | _native.NO_ZPAIR, _native.FORCE_TILED | _native.NO_MARCH,
                      _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED,
                      _native.FORCE_TILED | _native.NO_ZSEP | _native.NO_PACKED, _native.FORCE_DIRECT):
            got, info = run_case(vol, m, interp, flags)
            kernels.add(info.last_kernel)
            assert np.abs(got - ref)[mask].max() <= 2e-6, (interp, case, flags, info.last_kernel)
        assert {1, 2, 6} <= kernels and (case != 'rot_inplane' or ({3, 4, 5, 8} if LEGACY else {8}) <= kernels)


@pytest.mark.parametrize('interp', ['linear', 'bspline'])
def test_keep_outside_and_zero_fill(interp):
    """Outside voxels of a caller-supplied output: zero by default, untouched with keep_outside (the reference kernel's
    behaviour, transforms.py:276-278) -- on every kernel family, incl. the axis-exchanged marching paths."""
    shape = (40, 44, 48)
    vol = rand_vol(shape, 4)
    stale = np.full(shape, 7.0, dtype=np.float32)
    for mname in ('rot_general', 'rot_inplane45', 'rot_axis1_shift', 'rot_axis2_shift', 'shift_frac'):
        m = MATRICES[mname](shape)
        want_keep = oracle.affine(vol, m, interp, oracle.KEEP_OUTSIDE, output=stale.copy())
        want_zero = oracle.affine(vol, m, interp)
        assert (want_keep == 7.0).sum() > 100          # the case does have outside voxels
        for flags in (_native.FORCE_TILED, _native.FORCE_TILED | _native.FORCE_XSWAP, _native.FORCE_TILED | _native.FORCE_XSWAP | _native.NO_QUAD,
                      _native.FORCE_DIRECT):
            kept, _ = run_case(vol, m, interp, flags, keep=True, out_init=stale)
            zeroed, _ = run_case(vol, m, interp, flags, keep=False, out_init=stale)
            assert np.abs(kept - want_keep).max() <= 2e-6, (mname, flags)
            assert np.abs(zeroed - want_zero).max() <= 2e-6, (mname, flags)


def test_prefilter_matches_oracle_all_axes_lengths():
    """Prefilter alone: line lengths below/at/above the chunk and segment sizes, on every axis."""
    lib = _native.load()
    # (the last four: rows of whole 16-byte vectors -> the block-form strided passes; one segment, segments with warm-up at
    #  interior ends, partial last chunks, a last segment shorter than the warm-up)
    for shape in [(5, 7, 9), (12, 11, 13), (64, 65, 63), (130, 20, 70), (20, 200, 24), (24, 20, 300), (200, 17, 129),
                  (300, 260, 64), (100, 530, 32), (36, 257, 8), (270, 40, 260),
                  # rows of whole 8-sample lanes, lines of >= 40: X and Y fused in one kernel (prefilter_xy) -- one row segment, row
                  # segments with warm-up, a last segment shorter than the warm-up, column segments (W > 512), H at the tile edge
                  (3, 161, 64), (5, 200, 72), (2, 40, 520), (4, 300, 1000), (6, 128, 512), (3, 160, 96), (2, 45, 64), (3, 129, 136),
                  (2, 400, 992)]:
        vol = rand_vol(shape, 5)
        d = _native.DeviceArray.from_numpy(vol, 0)
        _native.check(lib.vt_prefilter_inplace(0, d.ptr, *shape), 'vt_prefilter_inplace')
        got = d.get()
        want = oracle.prefilter(vol)
        assert np.abs(got - want).max() <= 5e-6, shape
        d.free()


def test_prefilter_inplace_on_a_view_that_is_only_4_byte_aligned():
    """`vt_prefilter_inplace` on a caller's device pointer that is an offset view (4-byte aligned) of a shape the fused X+Y pass would
    otherwise take: the library must fall back to the separate passes, not fail."""
    import ctypes
    lib = _native.load()
    for shape in [(6, 128, 512), (3, 161, 64), (5, 200, 72)]:
        vol = rand_vol(shape, 23)
        n = int(np.prod(shape))
        buf = _native.DeviceArray((n + 8,), 0)
        for off in (1, 2, 3):                          # floats: 4-, 8-, 12-byte offsets from a 256-byte-aligned allocation
            host = np.zeros(n + 8, np.float32)
            host[off:off + n] = vol.ravel()
            buf.set(host)
            _native.check(lib.vt_prefilter_inplace(0, ctypes.c_void_p(buf.ptr + 4 * off), *shape), 'vt_prefilter_inplace')
            got = buf.get()
            assert np.abs(got[off:off + n].reshape(shape) - oracle.prefilter(vol)).max() <= 5e-6, (shape, off)
            assert not got[:off].any() and not got[off + n:].any()          # nothing outside the view was touched
        buf.free()


def test_packed_kernel_plain_tile_order_on_a_ragged_tile_count(monkeypatch):
    """VT_TILE_ORDER=0 (plain (d, h, w) tile order, an experiment knob) with a tile count that is no multiple of 64: every tile must
    still be computed (the per-XCD id partition rounded down and skipped the last tiles)."""
    monkeypatch.setenv('VT_TILE_ORDER', '0')
    for shape in ((72, 100, 130), (40, 136, 200)):
        vol = rand_vol(shape, 31)
        m = MATRICES['rot_general'](shape)
        want = oracle.affine(vol, m, 'linear')
        got, info = run_case(vol, m, 'linear', _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED)
        assert info.last_kernel == 6, info.last_kernel
        assert np.abs(got - want).max() <= TOL['linear'], shape


@pytest.mark.parametrize('shape', [(20, 47, 67), (9, 170, 101), (4, 290, 515), (3, 41, 64), (5, 161, 481)])
def test_prefilter_fused_xy_on_pitched_rows(shape):
    """The fused X+Y pass on the resident (pitched) layout, widths that are not multiples of 4 or 8: the identity transform of a
    filt_bspline volume returns the prefiltered-then-resampled samples; against the oracle on the whole volume."""
    vol = rand_vol(shape, 17)
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (0.25, -0.5, 0.125)                  # sub-voxel shift: every coefficient takes part
    sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu:0')
    got = sv.affine(m)
    want = oracle.affine(vol, m, 'filt_bspline')
    assert np.abs(got - want).max() <= TOL['filt_bspline'], shape
    sv.close()


def test_prefilter_known_answers():
    lib = _native.load()
    # constant volume: coefficients equal the constant away from the faces
    shape = (80, 72, 96)
    vol = np.full(shape, 0.75, dtype=np.float32)
    d = _native.DeviceArray.from_numpy(vol, 0)
    _native.check(lib.vt_prefilter_inplace(0, d.ptr, *shape), 'vt_prefilter_inplace')
    got = d.get()
    assert np.abs(got[16:-16, 16:-16, 16:-16] - 0.75).max() <= 2e-6
    d.free()
    # filt_bspline at integer positions reproduces the samples (interior)
    vol = rand_vol(shape, 6)
    sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu:0')
    got = sv.affine(np.eye(4, dtype=np.float32))
    assert np.abs(got - vol)[14:-14, 14:-14, 14:-14].max() <= 5e-6
    sv.close()


def test_known_answers_linear():
    shape = (64, 64, 64)
    vol = rand_vol(shape, 7)
    sv = vt.StaticVolume(vol, interpolation='linear', device='gpu:0')
    assert np.array_equal(sv.affine(np.eye(4, dtype=np.float32), _flags=_native.FORCE_TILED), vol)
    got = sv.translate((3, -2, 5), output=None)
    want = np.zeros_like(vol)
    want[3:, :-2, 5:] = vol[:-3, 2:, :-5]
    assert np.array_equal(got, want)
    sv.close()


def test_output_kinds_and_api_surface():
    shape = (48, 52, 56)
    vol = rand_vol(shape, 8)
    m = MATRICES['rot_scale_shift'](shape)
    want = oracle.affine(vol, m, 'linear')
    sv = vt.StaticVolume(vol, interpolation='linear', device='gpu')
    assert sv.shape == shape and sv.device == 'gpu' and sv.interpolation == 'linear'
    # numpy result
    assert np.abs(sv.affine(m) - want).max() <= 2e-6
    # DeviceArray output: returns None, result stays on the device
    buf = vt.zeros(shape, device='gpu:0')
    assert sv.affine(m, output=buf) is None
    assert np.abs(buf.get() - want).max() <= 2e-6
    # float64 matrix entry point
    buf.fill_zero()
    sv.affine(np.asarray(m, dtype=np.float64), output=buf)
    assert np.abs(buf.get() - want).max() <= 2e-6
    # front ends
    got = vt.transform(vol, rotation=(0, 30, 0), scale=1.2, interpolation='filt_bspline', device='gpu')
    c = centre(shape)
    m2 = vt.utils.transform_matrix(scale=(1.2, 1.2, 1.2), rotation=(0, 30, 0), center=c)
    assert np.abs(got - oracle.affine(vol, m2, 'filt_bspline')).max() <= 1e-5
    assert vt.rotate(vol, (10, 20, 30), device='gpu:0').shape == shape
    with pytest.raises(ValueError):
        vt.affine(vol, m, interpolation='nearest', device='gpu')
    with pytest.raises(ValueError):
        vt.StaticVolume(vol[0], device='gpu')
    sv.close()


def test_torch_tensor_output_and_input():
    torch = pytest.importorskip('torch')
    shape = (40, 40, 40)
    vol = rand_vol(shape, 9)
    m = MATRICES['rot_inplane45'](shape)
    out = torch.zeros(shape, dtype=torch.float32, device='cuda:0')
    sv = vt.StaticVolume(torch.from_numpy(vol).to('cuda:0'), interpolation='bspline', device='gpu:0')
    assert sv.affine(m, output=out) is None
    sv.synchronize()
    assert np.abs(out.cpu().numpy() - oracle.affine(vol, m, 'bspline')).max() <= 2e-6
    sv.close()


def test_reshape_gpu_matches_cpu_interior(golden_volumes, golden_volume):
    m = golden_volumes['reshape/matrix']
    ref = golden_volumes['reshape/linear']
    got = vt.affine(golden_volume, m, interpolation='linear', reshape=True, device='gpu')
    assert got.shape == ref.shape
    pad_before = golden_volumes['reshape/pad_before']
    m_eff = np.asarray(m, np.float64) @ np.asarray(vt.utils.translation_matrix(pad_before, np.float64))
    mask = interior_mask(m_eff, ref.shape, golden_volume.shape, 0)
    assert mask.sum() > 1000
    assert np.abs(got - ref)[mask].max() <= 2e-6


@pytest.mark.parametrize('interp', ALL_INTERPS)
@pytest.mark.parametrize('shape', [(40, 44, 48), (33, 47, 50)])
def test_projection_matches_oracle(interp, shape):
    """sum(axis=0) of the transformed volume: fused path for axis-0-separable matrices (kernel 7),
    transform + plane sum for the rest; both against the oracle's transformed volume summed in float64."""
    vol = rand_vol(shape, 7)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    cases = {k: MATRICES[k](shape) for k in ('identity', 'shift_int', 'shift_frac', 'rot_inplane45', 'rot_general', 'shear',
                                             'far_outside', 'minify')}
    cases['z_shift_frac'] = vt.utils.translation_matrix((2.5, 0.25, -1.5))
    cases['z_shift_neg'] = vt.utils.translation_matrix((-3.25, 0, 0))
    cases['z_shift_out'] = vt.utils.translation_matrix((shape[0] + 5.0, 0, 0))
    cases['z_shift_edge'] = vt.utils.translation_matrix((shape[0] - 0.75, 0, 0))
    tol = TOL[interp] * shape[0]
    for name, m in cases.items():
        want = oracle.affine(vol, m, interp).astype(np.float64).sum(axis=0)
        got = sv.projection(m)
        k = sv.info().last_kernel
        assert got.shape == shape[1:] and got.dtype == np.float32
        assert np.abs(got - want).max() <= tol, (interp, name, k, np.abs(got - want).max())
        separable = name in ('identity', 'shift_int', 'shift_frac', 'rot_inplane45', 'far_outside') or name.startswith('z_')
        assert (k == 7) == separable, (name, k)
        # the unfused path gives the same answer
        got2 = sv.projection(m, _flags=_native.NO_ZSEP)
        assert sv.info().last_kernel != 7
        assert np.abs(got2 - want).max() <= tol, (interp, name, 'unfused')
    # device output, and the keyword form
    out = vt.empty(shape[1:], device='gpu:0')
    assert sv.project(rotation=(30, 0, 0), rotation_order='sxyz', output=out) is None
    m = vt.utils.transform_matrix(rotation=(30, 0, 0), rotation_order='sxyz', center=centre(shape))
    want = oracle.affine(vol, m, interp).astype(np.float64).sum(axis=0)
    assert np.abs(out.get() - want).max() <= tol
    out.free()
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'bspline_simple', 'filt_bspline'])
def test_affine_batch_matches_single_calls(interp):
    """vt_volume_affine_batch: one launch for small volumes, queued launches for larger ones; host and device outputs."""
    for shape, n in (((20, 24, 28), 7), ((100, 100, 104), 3)):
        vol = rand_vol(shape, 9)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        rs = np.random.RandomState(3)
        ms = np.stack([vt.utils.transform_matrix(rotation=tuple(rs.uniform(-180, 180, 3)), rotation_order='sxyz',
                                                 translation=tuple(rs.uniform(-2, 2, 3)), center=centre(shape)) for _ in range(n)])
        ms[1] = np.eye(4)
        want = np.stack([sv.affine(m) for m in ms])
        got = sv.affine_batch(ms)
        assert got.shape == (n,) + shape and np.array_equal(got, want)
        out = vt.empty((n,) + shape, device='gpu:0')
        assert sv.affine_batch(ms, output=out) is None
        sv.synchronize()
        assert np.array_equal(out.get(), want)
        out.free()
        if shape[0] < 50:
            assert np.abs(got[0] - oracle.affine(vol, ms[0], interp)).max() <= TOL[interp]
            with pytest.raises(ValueError):
                sv.affine_batch(np.eye(4))
        sv.close()


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline'])
def test_oneshot_matches_resident_path(interp):
    """transform() on a host volume (vt_affine_oneshot: upload, prefilter, transform, download) against the resident
    StaticVolume path, which the other tests pin to the oracle; volume above the pinning threshold, results from the pool."""
    shape = (256, 192, 320)                       # 63 MB
    vol = rand_vol(shape, 21)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    # the pipelined one-shot (axis-0-separable matrices) samples the plain resident layout slab by slab, the resident handle
    # may pick the plane-pair / transposed copies: same arithmetic per tap, different summation order in the last bits
    tol = TOL[interp]
    cases = [vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(2.5, 1.0, -3.0), center=centre(shape)),
             vt.utils.translation_matrix((-7.25, 0.5, 0.0)),
             np.eye(4, dtype=np.float32),
             vt.utils.transform_matrix(rotation=(0, 100, 0), scale=(1.0, 1.2, 0.8), center=centre(shape)),
             vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=centre(shape)),
             vt.utils.translation_matrix((300.0, 0, 0)),
             vt.utils.translation_matrix((20.5, 0, 0)), vt.utils.translation_matrix((-40.0, 3, 0))]
    for i, m in enumerate(cases):
        want = sv.affine(m)
        got = vt.affine(vol, m, interpolation=interp, device='gpu')
        assert got.shape == shape
        assert np.abs(got - want).max() <= tol, (interp, i, float(np.abs(got - want).max()))
    # caller-supplied host output, and the reference's outside-voxel behaviour
    m = cases[0]
    out = np.full(shape, 7.0, dtype=np.float32)
    assert vt.affine(vol, m, interpolation=interp, device='gpu', output=out) is None
    assert np.abs(out - sv.affine(m)).max() <= tol
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline'])
def test_oneshot_pipeline_thin_volume(interp):
    """Pipelined one-shot on a thin volume: 8 chunks of 5 planes, one axis-0 prefilter chunk (it can only run when every plane
    is there), output slabs that need planes of several upload chunks."""
    shape = (40, 512, 520)                        # 42.6 MB
    vol = rand_vol(shape, 29)
    for m in (vt.utils.transform_matrix(rotation=(0, -50, 0), translation=(7.5, 3.0, -2.0), center=centre(shape)),
              vt.utils.translation_matrix((-12.25, 0.0, 0.0))):
        got = vt.affine(vol, m, interpolation=interp, device='gpu')
        assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp], interp


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline', 'filt_bspline_simple'])
def test_oneshot_pipeline_ragged_chunks(interp):
    """The pipelined one-shot (chunked upload, per-chunk prefilter passes, per-slab transform, chunked download) on a depth
    that the chunk count does not divide and that has a short last prefilter chunk, against the oracle; translations that
    make the first / last output slabs depend on far-away or on no source planes."""
    shape = (100, 300, 290)                       # 34.8 MB: above the pipeline threshold; 8 chunks of 13 planes, the last of 9
    vol = rand_vol(shape, 23)
    for m in (vt.utils.transform_matrix(rotation=(0, 20, 0), translation=(3.25, -1.0, 2.0), center=centre(shape)),
              vt.utils.translation_matrix((-30.5, 0.0, 1.0)), vt.utils.translation_matrix((61.0, 2.5, 0.0)),
              vt.utils.translation_matrix((150.0, 0.0, 0.0))):
        got = vt.affine(vol, m, interpolation=interp, device='gpu')
        assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp], interp
    # result written over the input: must not be pipelined (slabs would land on planes not yet uploaded)
    m = vt.utils.translation_matrix((-30.5, 0.0, 1.0))
    want = oracle.affine(vol, m, interp)
    inout = vol.copy()
    vt.affine(inout, m, interpolation=interp, device='gpu', output=inout)
    assert np.abs(inout - want).max() <= TOL[interp], interp


@pytest.mark.parametrize('interp', ['filt_bspline'])
def test_oneshot_pipeline_general_matrix(interp):
    """A general 3-D rotation through the pipelined one-shot call: every output slab waits for the whole (prefiltered) source, the
    plane-local prefilter passes run per uploaded chunk, the downloads overlap the later slabs' kernels.  Against the oracle and
    against the plain sequence (VT_NO_MARCH keeps a call off the pipeline)."""
    shape = (150, 700, 650)                       # 273 MB: above the general-matrix pipeline threshold (256 MiB), ragged chunks
    vol = rand_vol(shape, 31)
    for m in (vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', translation=(1.5, -2.0, 0.25), center=centre(shape)),
              vt.utils.transform_matrix(rotation=(0, 30, 0), rotation_order='sxyz', center=centre(shape))):      # about axis 1
        got = vt.affine(vol, m, interpolation=interp, device='gpu')
        assert np.abs(got - oracle.affine(vol, m, interp)).max() <= TOL[interp], interp


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline'])
def test_output_shape_other_than_source_shape(interp):
    """vt_volume_set_output_shape (scipy's output_shape, transforms.py:136-150): every kernel family with an output grid
    larger / smaller than the resident source, against the oracle's generalised entry point."""
    import ctypes
    lib = _native.load()
    shape = (40, 44, 48)
    vol = rand_vol(shape, 13)
    src = oracle.prefilter(vol) if interp.startswith('filt') else vol
    okind = 'bspline' if interp.startswith('filt') else interp
    for out_shape in ((50, 60, 72), (33, 30, 40)):
        h = ctypes.c_void_p()
        _native.check(lib.vt_volume_create(0, *shape, _native.INTERP_CODES[interp], vol.ctypes.data, 0, ctypes.byref(h)), 'create')
        _native.check(lib.vt_volume_set_output_shape(h, *out_shape), 'set_output_shape')
        c = centre(shape)
        for mname, m in (('axis0', vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, -2, 3), center=c)),
                         ('axis0_100', vt.utils.transform_matrix(rotation=(0, 100, 0), center=c)),
                         ('axis1', vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='sxyz', center=c)),
                         ('axis2', vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', center=c)),
                         ('general', vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', scale=(1.2, 0.9, 1.1), center=c))):
            m32 = np.ascontiguousarray(m, dtype=np.float32)
            want = oracle.affine_ex(src, np.asarray(m32, np.float64), okind, out_shape)
            for flags in (0, _native.FORCE_TILED | _native.FORCE_XSWAP, _native.FORCE_TILED | _native.FORCE_XSWAP | _native.NO_QUAD,
                          _native.FORCE_TILED | _native.NO_ZPAIR, _native.FORCE_TILED | _native.NO_ZSEP,
                          _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED, _native.FORCE_DIRECT):
                got = np.empty(out_shape, np.float32)
                _native.check(lib.vt_volume_affine(h, m32.ctypes.data, got.ctypes.data, flags), 'affine')
                assert np.abs(got - want).max() <= TOL[interp], (interp, out_shape, mname, flags)
            pr = np.empty(out_shape[1:], np.float32)
            _native.check(lib.vt_volume_project(h, m32.ctypes.data, pr.ctypes.data, 0), 'project')
            assert np.abs(pr - want.astype(np.float64).sum(axis=0)).max() <= TOL[interp] * out_shape[0], (interp, out_shape, mname, 'projection')
        lib.vt_volume_destroy(h)


@pytest.mark.parametrize('interp', ['linear', 'bspline'])
def test_projection_repeated_calls_large_plane(interp):
    """A plane large enough for the tiled kernels; successive projections must not see stale helper state."""
    shape = (5, 512, 520)
    vol = rand_vol(shape, 8)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for shift, ang in ((0.0, 10.0), (1.5, -35.0), (-0.75, 80.0)):
        m = vt.utils.transform_matrix(rotation=(ang, 0, 0), rotation_order='sxyz', translation=(shift, 2.0, -3.5),
                                      center=centre(shape))
        want = oracle.affine(vol, m, interp).astype(np.float64).sum(axis=0)
        got = sv.projection(m)
        assert sv.info().last_kernel == 7
        assert np.abs(got - want).max() <= TOL[interp] * shape[0], (interp, shift, ang)
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline_simple'])
def test_projection_tilt_series_reuses_the_plane_sum(interp, monkeypatch):
    """The reference's tilt series (examples/projections.py:20-26: rotation about the projection axis, then sum(axis=0)) asks for
    the same weighted plane sum at every angle: the handle keeps it and only the 2-D interpolation runs again.  Every projection of
    the series must equal the oracle's transform-then-sum and the uncached path bit for bit, also after the axis-0 offset or the
    output depth changed in between (the two things the sum depends on)."""
    shape = (40, 120, 136)
    vol = rand_vol(shape, 21)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    series = [(0.0, a) for a in (-60.0, -21.0, 3.0, 45.0)] + [(1.25, 45.0), (1.25, -10.0), (0.0, -10.0), (-2.0, 30.0)]
    got = []
    for shift, ang in series:
        m = vt.utils.transform_matrix(rotation=(ang, 0, 0), rotation_order='sxyz', translation=(shift, 1.0, -2.5), center=centre(shape))
        got.append(sv.projection(m))
        assert sv.info().last_kernel == 7
        want = oracle.affine(vol, m, interp).astype(np.float64).sum(axis=0)
        assert np.abs(got[-1] - want).max() <= TOL[interp] * shape[0], (interp, shift, ang)
    sv.close()
    monkeypatch.setenv('VT_NO_PROJ_CACHE', '1')                    # (a handle reads its knobs when it is created)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for (shift, ang), g in zip(series, got):
        m = vt.utils.transform_matrix(rotation=(ang, 0, 0), rotation_order='sxyz', translation=(shift, 1.0, -2.5), center=centre(shape))
        assert np.array_equal(sv.projection(m), g), (interp, shift, ang)
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
def test_full_size_properties_512(interp):
    """BASELINE sizes: properties that need no CPU oracle pass over 134M voxels."""
    n = 512
    rs = np.random.RandomState(10)
    vol = rs.random_sample((n, n, n)).astype(np.float32)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    out = vt.empty((n, n, n), device='gpu:0')
    tol = TOL[interp]
    # identity: linear returns the input bit-for-bit; filt_bspline reproduces it in the interior
    sv.affine(np.eye(4, dtype=np.float32), output=out)
    got = out.get()
    assert sv.info().last_kernel == 8
    if interp == 'linear':
        assert np.array_equal(got, vol)
    else:
        assert np.abs(got - vol)[14:-14, 14:-14, 14:-14].max() <= 5e-6
    # integer shift is a shifted copy with zero fill (linear) -- no resampling error
    if interp == 'linear':
        sv.translate((7, -3, 11), output=out)
        got = out.get()
        want = np.zeros_like(vol)
        want[7:, :-3, 11:] = vol[:-7, 3:, :-11]
        assert np.array_equal(got, want)
    # a rotated slab of the big volume equals the oracle on a sub-block (oracle finishes in seconds on 24 planes)
    m = vt.utils.transform_matrix(rotation=(0, 45, 0), center=centre((n, n, n)))
    sv.affine(m, output=out)
    got = out.get()
    assert sv.info().last_kernel == 8
    d0 = 200
    if interp == 'linear':
        want = oracle.affine_ex(vol, np.asarray(m, np.float64), 'linear', (8, n, n), out_plane0=d0)
    else:
        # the headline kernel (pair marching, round-aware chunks of 172 planes) against the oracle: prefilter a window of
        # +-40 planes around the sub-block (|z|^38 = 2e-22 of the cut reaches the tapped planes), then sample it
        w0, w1 = d0 - 40, d0 + 8 + 40
        want = oracle.affine_ex(oracle.prefilter(vol[w0:w1]), np.asarray(m, np.float64), 'bspline', (8, n, n),
                                plane0=w0, global_depth=n, out_plane0=d0)
    assert np.abs(got[d0:d0 + 8] - want).max() <= tol
    # general rotation: tiled result equals the direct-gather result (two independent kernels)
    m = vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=centre((n, n, n)))
    sv.affine(m, output=out)
    a = out.get()
    assert sv.info().last_kernel == (6 if interp == 'linear' else 9)     # trilinear: packed footprints; cubic: lane blocks
    sv.affine(m, output=out, _flags=_native.FORCE_DIRECT)
    b = out.get()
    assert np.abs(a - b).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_BLOCK)
    assert sv.info().last_kernel == (6 if interp == 'linear' else 2)      # packed footprints are planned for trilinear only
    assert np.abs(out.get() - b).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_PACKED)
    assert sv.info().last_kernel == 2
    assert np.abs(out.get() - b).max() <= tol
    # rotation about axis 1 / 2: marching on the axis-exchanged copy vs the general kernel vs the direct kernel
    for rot in ((0, 33, 0), (0, 0, 33)):
        m = vt.utils.transform_matrix(rotation=rot, rotation_order='sxyz', translation=(0.25, 1.5, -2.0), center=centre((n, n, n)))
        sv.affine(m, output=out)
        a = out.get()
        assert sv.info().last_kernel == (10 if rot[2] else 8), rot      # about array axis 2: the row kernel, whatever the offsets (round 5)
        sv.affine(m, output=out, _flags=_native.NO_ZSEP)
        assert sv.info().last_kernel == (9 if interp != 'linear' else sv.info().last_kernel)
        assert np.abs(a - out.get()).max() <= tol, rot
        sv.affine(m, output=out, _flags=_native.NO_ZSEP | _native.NO_BLOCK)
        assert sv.info().last_kernel in (2, 6)
        assert np.abs(a - out.get()).max() <= tol, rot
        sv.affine(m, output=out, _flags=_native.FORCE_DIRECT)
        assert np.abs(a - out.get()).max() <= tol, rot
    # in-plane rotation: separable kernel vs general kernel vs direct kernel
    m = vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(0.25, 1.5, -2.0), center=centre((n, n, n)))
    sv.affine(m, output=out)
    a = out.get()
    assert sv.info().last_kernel == 8
    sv.affine(m, output=out, _flags=_native.NO_QUAD)
    assert sv.info().last_kernel in (((4,) if interp == 'linear' else (5,)) if LEGACY else GENERAL_KERNELS)
    assert np.abs(a - out.get()).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_ZPAIR)
    assert sv.info().last_kernel in ((4,) if LEGACY else GENERAL_KERNELS)
    assert np.abs(a - out.get()).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_MARCH)
    assert sv.info().last_kernel in ((3,) if LEGACY else GENERAL_KERNELS)
    assert np.abs(a - out.get()).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_ZSEP)
    assert sv.info().last_kernel == (6 if interp == 'linear' else 9)     # a source beyond the memory-side cache: footprints, not boxes
    assert np.abs(a - out.get()).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_ZSEP | _native.NO_BLOCK)
    assert sv.info().last_kernel == (6 if interp == 'linear' else 2)
    assert np.abs(a - out.get()).max() <= tol
    sv.affine(m, output=out, _flags=_native.NO_ZSEP | _native.FORCE_PACKED)
    assert sv.info().last_kernel == 6
    assert np.abs(a - out.get()).max() <= tol
    sv.affine(m, output=out, _flags=_native.FORCE_DIRECT)
    assert np.abs(a - out.get()).max() <= tol
    sv.close()
    out.free()


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline'])
def test_slab_handles_reproduce_whole_volume(interp):
    """vt_volume_create_slab: resident windows with halo planes + output-plane offsets (the multi-GPU building block),
    all slabs on one GPU here."""
    import ctypes
    from voltools_amd.distributed import plan_halo_exchange, slab_bounds, stencil_halo
    lib = _native.load()
    counts = [30, 36, 30]
    G, H, W = sum(counts), 40, 46
    vol = rand_vol((G, H, W), 11)
    c = centre((G, H, W))
    mats = [vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(0.0, 1.5, -2.0), center=c),
            vt.utils.transform_matrix(rotation=(0, 45, 0), translation=(1.25, 0, 0), center=c)]
    halo = stencil_halo(interp) + 2
    tol = TOL[interp]        # filt_*: the window carries 16 warm-up planes (|z|^16 = 7e-10): same bar as a whole resident volume
    for m in mats:
        want = oracle.affine(vol, m, interp)
        projs = {}
        for flags in (0, _native.NO_QUAD, _native.NO_ZPAIR, _native.NO_MARCH, _native.NO_ZSEP, _native.FORCE_DIRECT):
            parts = []
            for r, (g0, g1) in enumerate(slab_bounds(counts)):
                (w0, w1), _, _ = plan_halo_exchange(counts, r, halo)
                win = np.ascontiguousarray(vol[w0:w1])
                cflags = (_native.SLAB_LO_INTERIOR if w0 > 0 else 0) | (_native.SLAB_HI_INTERIOR if w1 < G else 0)
                h = ctypes.c_void_p()
                _native.check(lib.vt_volume_create_slab(0, w1 - w0, H, W, _native.INTERP_CODES[interp], win.ctypes.data,
                                                        cflags, w0, G, g0, g1 - g0, ctypes.byref(h)), 'vt_volume_create_slab')
                out = np.empty((g1 - g0, H, W), np.float32)
                m32 = np.ascontiguousarray(m, dtype=np.float32)
                _native.check(lib.vt_volume_affine(h, m32.ctypes.data, out.ctypes.data, flags | _native.FORCE_TILED * (flags != _native.FORCE_DIRECT)),
                              'vt_volume_affine')
                # the slab's share of the axis-0 projection (fused for flags 0, transform + sum with NO_ZSEP)
                if flags in (0, _native.NO_ZSEP):
                    pr = np.empty((H, W), np.float32)
                    _native.check(lib.vt_volume_project(h, m32.ctypes.data, pr.ctypes.data, flags), 'vt_volume_project')
                    projs.setdefault(flags, []).append(pr.astype(np.float64))
                lib.vt_volume_destroy(h)
                parts.append(out)
            got = np.concatenate(parts)
            assert np.abs(got - want).max() <= tol, (interp, flags)
        for flags, prs in projs.items():
            assert np.abs(sum(prs) - want.astype(np.float64).sum(axis=0)).max() <= tol * G, (interp, flags, 'projection')


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
def test_deferred_handle_upload_planes_finalize(interp):
    """VT_SRC_DEFERRED / vt_volume_upload_planes / vt_volume_finalize (the construction path of SlabVolume): a window filled plane
    range by plane range, from host and from device memory, equals the handle built from the assembled window; planes never
    uploaded read as zeros; the handle refuses transforms before it is finalized and uploads after."""
    import ctypes
    lib = _native.load()
    D, H, W = 44, 40, 50
    vol = rand_vol((D, H, W), 31)
    m = np.ascontiguousarray(vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(0.5, 1.5, -2.0), center=centre((D, H, W))), dtype=np.float32)
    code = _native.INTERP_CODES[interp]

    def run(handle):
        out = np.empty((D, H, W), np.float32)
        _native.check(lib.vt_volume_affine(handle, m.ctypes.data, out.ctypes.data, _native.FORCE_TILED), 'affine')
        return out

    ref_h = ctypes.c_void_p()
    _native.check(lib.vt_volume_create(0, D, H, W, code, vol.ctypes.data, 0, ctypes.byref(ref_h)), 'create')
    want = run(ref_h)
    lib.vt_volume_destroy(ref_h)

    h = ctypes.c_void_p()
    _native.check(lib.vt_volume_create_slab(0, D, H, W, code, None, _native.SRC_DEFERRED, 0, D, 0, D, ctypes.byref(h)), 'create deferred')
    out = np.empty((D, H, W), np.float32)
    assert lib.vt_volume_affine(h, m.ctypes.data, out.ctypes.data, 0) != 0            # not finalized yet
    a = np.ascontiguousarray(vol[10:30])
    _native.check(lib.vt_volume_upload_planes(h, 10, 20, a.ctypes.data, 0), 'upload host')
    dev = _native.DeviceArray.from_numpy(np.ascontiguousarray(vol[:10]), 0)
    _native.check(lib.vt_volume_upload_planes(h, 0, 10, ctypes.c_void_p(dev.ptr), _native.SRC_DEVICE), 'upload device')
    b = np.ascontiguousarray(vol[30:])
    _native.check(lib.vt_volume_upload_planes(h, 30, D - 30, b.ctypes.data, 0), 'upload host 2')
    assert lib.vt_volume_upload_planes(h, 40, 10, b.ctypes.data, 0) != 0               # beyond the window
    _native.check(lib.vt_volume_finalize(h), 'finalize')
    assert lib.vt_volume_upload_planes(h, 0, 1, a.ctypes.data, 0) != 0                 # finalized: no more uploads
    got = run(h)
    lib.vt_volume_destroy(h)
    dev.free()
    assert np.array_equal(got, want)

    # planes that were never uploaded are zeros
    h = ctypes.c_void_p()
    _native.check(lib.vt_volume_create_slab(0, D, H, W, code, None, _native.SRC_DEFERRED, 0, D, 0, D, ctypes.byref(h)), 'create deferred')
    _native.check(lib.vt_volume_upload_planes(h, 10, 20, a.ctypes.data, 0), 'upload host')
    _native.check(lib.vt_volume_finalize(h), 'finalize')
    got = run(h)
    lib.vt_volume_destroy(h)
    holes = vol.copy()
    holes[:10] = 0
    holes[30:] = 0
    assert np.abs(got - oracle.affine(holes, m, interp)).max() <= TOL[interp]


@pytest.mark.timeout(300, method='thread')     # an RCCL bring-up that hangs on a bad box must not hold the whole run
def test_slab_volume_single_rank_process_group():
    """The product multi-GPU class end to end on one rank (RCCL process group of size 1).  Kept last in this file."""
    torch = pytest.importorskip('torch')
    import torch.distributed as dist
    from voltools_amd.distributed import SlabVolume
    import os
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        shape = (64, 48, 52)
        vol = rand_vol(shape, 12)
        sv = SlabVolume(vol, interpolation='filt_bspline', device='gpu:0')
        m = vt.utils.transform_matrix(rotation=(0, 45, 0), center=centre(shape))
        out = vt.empty(shape, device='gpu:0')
        assert sv.affine(m, output=out) is None
        sv.synchronize()
        assert np.abs(out.get() - oracle.affine(vol, m, 'filt_bspline')).max() <= TOL['filt_bspline']
        # a single slab holds the whole volume: any matrix is within reach
        m = vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=centre(shape))
        got = sv.affine(m)
        assert np.abs(got - oracle.affine(vol, m, 'filt_bspline')).max() <= TOL['filt_bspline']
        # projection through the slab handle + the (size-1) all-reduce
        m = vt.utils.transform_matrix(rotation=(0, 30, 0), translation=(1.25, 0, 0), center=centre(shape))
        proj = sv.projection(m)
        assert proj.is_cuda and tuple(proj.shape) == shape[1:]
        want = oracle.affine(vol, m, 'filt_bspline').astype(np.float64).sum(axis=0)
        assert np.abs(proj.cpu().numpy() - want).max() <= TOL['filt_bspline'] * shape[0]
        sv.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('interp', ['linear', 'bspline_simple'])
def test_lane_block_kernel_all_paths(interp, monkeypatch):
    """Kernel 9 on shapes with partial tiles, boxes that leave the volume (checked staging), keep_outside and -- through the
    VT_BLOCK_LINEAR switch, read at create -- its trilinear instantiation, which the planner does not pick by itself."""
    monkeypatch.setenv('VT_BLOCK_LINEAR', '1')
    for shape in ((70, 66, 72), (33, 47, 50), (130, 40, 97)):
        vol = rand_vol(shape, 3)
        for mname in ('rot_general', 'rot_scale_shift', 'shear', 'mirror', 'rot_axis2_shift'):
            m = MATRICES[mname](shape)
            want = oracle.affine(vol, m, interp)
            got, info = run_case(vol, m, interp, _native.FORCE_TILED | _native.NO_ZSEP)
            assert info.last_kernel == 9, (shape, mname, info.last_kernel)
            assert np.abs(got - want).max() <= TOL[interp], (shape, mname)
            init = rand_vol(shape, 4)
            got, info = run_case(vol, m, interp, _native.FORCE_TILED | _native.NO_ZSEP, keep=True, out_init=init)
            assert info.last_kernel == 9
            assert np.abs(got - oracle.affine(vol, m, interp, oracle.KEEP_OUTSIDE, output=init.copy())).max() <= TOL[interp], (shape, mname, 'keep')


@pytest.mark.parametrize('interp', ['bspline', 'bspline_simple', 'filt_bspline', 'filt_bspline_simple'])
@pytest.mark.parametrize('shape', [(70, 66, 72), (5, 40, 48), (33, 130, 70), (2, 64, 64)])
def test_zconvolved_copy_against_the_oracle_and_the_four_plane_kernel(interp, shape):
    """Cubic plane-quad launches with an integer axis-0 offset sample the z-convolved copy (KIND 4: the axis-0 weights of such a launch are
    the constants 1/6, 2/3, 1/6, 0, applied once when the copy is built).  Against the oracle at the family's tolerance, and against the
    four-tap-plane kernel (VT_NO_ZFIR) on the same handle: same taps, one more rounding of the axis-0 sum, so 2 ulps of the data range.
    Depths that are not multiples of four, volumes thinner than the stencil, offsets that push the stencil over either end of axis 0, and
    the copies of the exchanged orientations (rotations about axes 1 / 2)."""
    vol = rand_vol(shape, 21)
    c = centre(shape)
    cases = {
        'rot33': vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='rzxz', center=c),
        'rot100_up2': vt.utils.transform_matrix(rotation=(0, 100, 0), rotation_order='rzxz', translation=(2, 0.25, -1.5), center=c),
        'rot33_down3': vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='rzxz', translation=(-3, 0, 0), center=c),
        'shift_inplane': vt.utils.translation_matrix((0, 0.5, -1.25)),
        'shift_past_the_end': vt.utils.translation_matrix((shape[0] - 1, 0.5, 0.25)),
        'scale_inplane': vt.utils.transform_matrix(scale=(1.0, 1.2, 0.8), center=c),
        'rot_axis1': vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='sxyz', center=c),
        'rot_axis2': vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', center=c),
    }
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    base = sv.info().resident_bytes
    used_quad = 0
    for name, m in cases.items():
        want = oracle.affine(vol, m, interp)
        got = sv.affine(m, _flags=_native.FORCE_TILED)
        k = sv.info().last_kernel
        four = sv.affine(m, _flags=_native.FORCE_TILED | _native.NO_ZFIR)
        assert sv.info().last_kernel == k, name
        assert np.abs(got - want).max() <= TOL[interp], (interp, shape, name, k)
        assert np.abs(four - want).max() <= TOL[interp], (interp, shape, name, k)
        assert np.abs(got - four).max() <= 4.8e-7 * (3.0 if interp.startswith('filt') else 1.0), (interp, shape, name, k)
        used_quad += int(k == 8)
    if used_quad:
        # both copies of every orientation that marched are accounted for
        assert sv.info().resident_bytes >= base + 2 * vol.nbytes
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
def test_release_copies_frees_the_lazy_copies_and_changes_no_result(interp):
    """`vt_volume_release_copies`: after rotations about all three axes a handle holds the exchanged orientations and their plane-quad forms
    (several times the volume); releasing them returns the handle to its plain copy, and the next calls rebuild what they need and return
    the same bits."""
    shape = (96, 88, 104)
    vol = rand_vol(shape, 31)
    c = centre(shape)
    mats = [vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='rzxz', center=c),
            vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='rzxz', translation=(0.5, 0, 0), center=c),
            vt.utils.transform_matrix(rotation=(0, 33, 0), rotation_order='sxyz', center=c),
            vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', center=c),
            vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=c)]
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    base = sv.info().resident_bytes
    first = [sv.affine(m, _flags=_native.FORCE_TILED) for m in mats]
    grown = sv.info().resident_bytes
    assert grown >= base + 3 * vol.nbytes                     # at least the exchanged copies and the plane-quad forms of the sweeps
    freed = sv.release_copies()
    assert freed == grown - sv.info().resident_bytes and sv.info().resident_bytes == base
    assert sv.release_copies() == 0                           # idempotent
    for m, want in zip(mats, first):
        assert np.array_equal(sv.affine(m, _flags=_native.FORCE_TILED), want)
        assert np.abs(want - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    assert sv.info().resident_bytes == grown
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'bspline', 'filt_bspline_simple'])
def test_general_matrices_on_the_copy_whose_rows_follow_the_output_w_axis(interp):
    """General matrices whose output w direction follows source axis 0 / 1 sample the axis-permuted resident copies (`try_general_reorient`:
    same taps, the source side of the launch permuted).  Every general-matrix kernel family against the oracle at the family's tolerance
    and against the same family on the plain copy (`VT_NO_REORIENT`): the two differ by the order of the per-axis sums only."""
    shape = (70, 66, 72)
    vol = rand_vol(shape, 41)
    c = centre(shape)
    cases = {
        'w_follows_axis0': vt.utils.transform_matrix(rotation=(20, 75, 10), rotation_order='sxyz', center=c),
        'w_follows_axis1': vt.utils.transform_matrix(rotation=(80, 10, 15), rotation_order='sxyz', center=c),
        'w_follows_axis0_affine': vt.utils.transform_matrix(rotation=(-30, 110, 5), scale=(1.1, 0.9, 1.2), translation=(1.5, -2, 0.75), rotation_order='sxyz', center=c),
        'w_follows_axis1_mirror': vt.utils.transform_matrix(rotation=(-100, 12, 7), scale=(1.0, -1.0, 1.0), rotation_order='sxyz', center=c),
    }
    FT = _native.FORCE_TILED
    families = (FT, FT | _native.NO_BLOCK, FT | _native.NO_PACKED, FT | _native.FORCE_PACKED)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    base = sv.info().resident_bytes
    for name, m in cases.items():
        m64 = np.asarray(m, np.float64)
        follows = int(np.argmax(np.abs(m64[:3, 2])))
        assert follows == (0 if 'axis0' in name else 1), name
        want = oracle.affine(vol, m, interp)
        for flags in families:
            got = sv.affine(m, _flags=flags)
            k = sv.info().last_kernel
            assert k in (2, 6, 9)
            plain = sv.affine(m, _flags=flags | _native.NO_REORIENT)
            assert np.abs(got - want).max() <= TOL[interp], (interp, name, flags, k)
            assert np.abs(plain - want).max() <= TOL[interp], (interp, name, flags, k)
            assert np.abs(got - plain).max() <= 2 * TOL[interp]
    assert sv.info().resident_bytes >= base + 2 * vol.nbytes          # both permuted copies were built
    sv.close()


def test_reoriented_copies_are_built_at_the_fourth_request_only():
    """Without VT_FORCE_TILED a handle builds an axis-permuted copy when the FOURTH general matrix asks for it (a one-shot handle never
    pays the transpose pass), only from 192^3 outputs on, and the result is the same array either way to the family's tolerance."""
    shape = (200, 192, 208)
    vol = rand_vol(shape, 43)
    c = centre(shape)
    m = vt.utils.transform_matrix(rotation=(20, 75, 10), rotation_order='sxyz', center=c)
    sv = vt.StaticVolume(vol, interpolation='linear', device='gpu:0')
    base = sv.info().resident_bytes
    outs = []
    for i in range(5):
        outs.append(sv.affine(m))
        grown = sv.info().resident_bytes > base
        assert grown == (i >= 3), i
    assert np.abs(outs[0] - outs[4]).max() <= 2 * TOL['linear']
    assert np.array_equal(outs[3], outs[4])
    want = oracle.affine_ex(vol, np.asarray(m, np.float64), 'linear', (8,) + shape[1:], out_plane0=90)     # output planes 90..97
    assert np.abs(outs[4][90:98] - want).max() <= TOL['linear']
    assert np.abs(outs[0][90:98] - want).max() <= TOL['linear']
    small = vt.StaticVolume(vol[:100, :100, :100].copy(), interpolation='linear', device='gpu:0')
    b0 = small.info().resident_bytes
    ms = vt.utils.transform_matrix(rotation=(20, 75, 10), rotation_order='sxyz', center=centre((100, 100, 100)))
    for _ in range(6):
        small.affine(ms)
    assert small.info().resident_bytes == b0
    small.close()
    sv.close()


@pytest.mark.parametrize('cubic_frac', [False, True], ids=['default', 'VT_ROWS=2'])
@pytest.mark.parametrize('interp', ALL_INTERPS)
@pytest.mark.parametrize('shape', [(70, 66, 72), (33, 47, 50), (5, 9, 130), (64, 64, 64)])
def test_row_kernel_for_maps_that_leave_axis_2_alone(interp, shape, cubic_frac, monkeypatch):
    """Kind 10 (vt_kernels_rows.hip): rotations about axis 2 and any (d, h) affine map with any axis-2 offset.  Against the oracle at the
    family's tolerance and BIT-IDENTICAL to affine_direct (same chain of operations; for integer offsets the x-sum of the cubic stencil is
    formed once in the x-convolved copy); widths that are no multiple of 64 or 4, offsets that push rows over either end, keep_outside.
    Cubic launches with a FRACTIONAL offset take the row kernel only under VT_ROWS=2 (64 taps per voxel from the plain copy: measured
    slower than the exchange path, vt_plan.hip::plan_rows); the second parametrisation holds that form to the same bits."""
    if cubic_frac:
        if interp == 'linear':
            pytest.skip('the knob changes cubic launches only')
        monkeypatch.setenv('VT_ROWS', '2')
    slow_frac = interp != 'linear' and not cubic_frac
    vol = rand_vol(shape, 51)
    c = centre(shape)
    rot = vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', center=c)
    cases = {'rot33': (rot, True)}
    # (round 5: integers that are no multiple of four stage 18 vectors per row instead of 16; fractional offsets add the x taps of the
    #  interpolation -- trilinear always, cubic on request)
    for name, t2, takes in (('rot33_w+8', 8.0, True), ('rot33_w-4', -4.0, True), ('rot33_w+2', 2.0, True), ('rot33_w+0.5', 0.5, True),
                            ('rot33_w-3', -3.0, True), ('rot33_w+1.25', 1.25, True), ('rot33_w-7.75', -7.75, True), ('rot33_w+61', 61.0, True)):
        m = rot.copy(); m[2, 3] += t2
        cases[name] = (m, takes and not (slow_frac and t2 != np.floor(t2)))
    m = vt.utils.transform_matrix(rotation=(0, 0, -100), scale=(1.2, 0.8, 1.0), translation=(1.5, -2.25, 0.0), rotation_order='sxyz', center=c)
    cases['rot_scale_dh'] = (m, True)
    m = vt.utils.transform_matrix(rotation=(0, 0, 200), rotation_order='sxyz', center=c); m[2, 3] = float(shape[2] - 4 - (shape[2] % 4))
    cases['rot200_w_far'] = (m, True)
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    base = sv.info().resident_bytes
    for name, (m, takes) in cases.items():
        m64 = np.asarray(m, np.float64)
        assert m64[2, 0] == 0 and m64[2, 1] == 0 and m64[2, 2] == 1 and m64[0, 2] == 0 and m64[1, 2] == 0, name
        want = oracle.affine(vol, m, interp)
        got = sv.affine(m, _flags=_native.FORCE_TILED)
        k = sv.info().last_kernel
        assert (k == 10) == takes, (interp, shape, name, k)
        assert np.abs(got - want).max() <= TOL[interp], (interp, shape, name, k)
        direct = sv.affine(m, _flags=_native.FORCE_DIRECT)
        if takes:
            assert np.array_equal(got, direct), (interp, shape, name, float(np.abs(got - direct).max()))
            other = sv.affine(m, _flags=_native.FORCE_TILED | _native.NO_ROWS)
            assert sv.info().last_kernel != 10
            assert np.abs(other - want).max() <= TOL[interp]
            init = rand_vol(shape, 52)
            kept = init.copy()
            sv.affine(m, output=kept, keep_outside=True, _flags=_native.FORCE_TILED)
            assert sv.info().last_kernel == 10
            outside = (got == 0) & (direct == 0) & (np.abs(want) == 0)
            kd = init.copy()
            sv.affine(m, output=kd, keep_outside=True, _flags=_native.FORCE_DIRECT)
            assert np.array_equal(kept, kd), (interp, shape, name)
    if interp != 'linear':
        assert sv.info().resident_bytes >= base + vol.nbytes          # the x-convolved copy (integer offsets)
        assert sv.release_copies() > 0 and sv.info().resident_bytes == base
        assert np.array_equal(sv.affine(rot, _flags=_native.FORCE_TILED), sv.affine(rot, _flags=_native.FORCE_DIRECT))
    sv.close()


@pytest.mark.parametrize('interp', ['linear'])
def test_packed_span_kernel_forms_and_rim_paths(interp, monkeypatch):
    """Round 5's trilinear packed-span kernel (vt_kernels_span.hip, kind 6) in both forms -- single buffer (`affine_span`) and wave-specialised
    with two buffers (`affine_span_ws`, VT_SPAN_PIPE=1: the knob is read when a handle is created) -- on shapes that exercise what is new:
    rim tiles on every face (store masks ahead of the gather, bounds tests by dividing the staging offset), partial tiles at the output's
    far ends, exact-integer coordinates (quarter turns: the Q32.32 terms must round, not truncate), keep_outside, a reshaped output.
    Against the oracle at the family's tolerance; the two forms against each other bit for bit (same arithmetic per voxel)."""
    shapes = [(70, 66, 72), (130, 40, 64), (48, 200, 130), (97, 65, 200)]
    flags = _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED
    for shape in shapes:
        vol = rand_vol(shape, 77)
        c = centre(shape)
        mats = {
            'general': vt.utils.transform_matrix(rotation=(25.0, -40.0, 70.0), rotation_order='sxyz', translation=(1.5, -2.0, 0.75), center=c),
            'general_scaled': vt.utils.transform_matrix(rotation=(-110.0, 33.0, 12.0), scale=(0.8, 1.1, 1.3), rotation_order='rzxz', center=c),
            'quarter': vt.utils.transform_matrix(rotation=(90.0, 180.0, 270.0), rotation_order='sxyz', center=c),
            'quarter2': vt.utils.transform_matrix(rotation=(270.0, 90.0, 0.0), rotation_order='sxyz', center=c),
            'far_corner': vt.utils.transform_matrix(rotation=(10.0, 20.0, 30.0), rotation_order='sxyz', translation=(shape[0] * 0.4, -shape[1] * 0.3, 5.0), center=c),
        }
        got = {}
        for pipe in ('0', '1'):
            monkeypatch.setenv('VT_SPAN_PIPE', pipe)
            sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
            for name, m in mats.items():
                want = oracle.affine(vol, m, interp)
                out = sv.affine(m, _flags=flags)
                k = sv.info().last_kernel
                assert k in (6, 2), (shape, name, pipe, k)               # (tiny boxes: the span family refuses volumes smaller than its box)
                assert np.abs(out - want).max() <= TOL[interp], (shape, name, pipe, k, float(np.abs(out - want).max()))
                got[(name, pipe)] = (out, k)
                init = rand_vol(shape, 78)
                kept = init.copy()
                sv.affine(m, output=kept, keep_outside=True, _flags=flags)
                kd = init.copy()
                sv.affine(m, output=kd, keep_outside=True, _flags=_native.FORCE_DIRECT)
                assert np.array_equal(kept == init, kd == init) or np.abs(kept - kd).max() <= TOL[interp], (shape, name, pipe)
            sv.close()
        for name in mats:
            (a, ka), (b, kb) = got[(name, '0')], got[(name, '1')]
            if ka == 6 and kb == 6:
                assert np.array_equal(a, b), (shape, name, float(np.abs(a - b).max()))


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
def test_row_kernel_with_two_row_buffers_returns_the_same_bits(interp, monkeypatch):
    """`affine_rows_db` (VT_ROWS_DB=1: a workgroup walks all runs of its pixel tile with two row buffers, one barrier per run, a counted
    vmcnt; measured slower than the one-run form and therefore not the default) against the default form, bit for bit -- same arithmetic per
    voxel -- on widths with a ragged last run, offsets of every kind, keep_outside."""
    shape = (70, 66, 200)
    vol = rand_vol(shape, 61)
    c = centre(shape)
    rot = vt.utils.transform_matrix(rotation=(0, 0, 33), rotation_order='sxyz', center=c)
    mats = []
    for t2 in (0.0, 8.0, -3.0, 0.5, -7.75):
        m = rot.copy(); m[2, 3] += t2
        mats.append(m)
    got = {}
    monkeypatch.setenv('VT_ROWS', '2')                      # (cubic launches with fractional offsets on the row kernel too)
    for db in ('0', '1'):
        monkeypatch.setenv('VT_ROWS_DB', db)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for i, m in enumerate(mats):
            out = sv.affine(m, _flags=_native.FORCE_TILED)
            assert sv.info().last_kernel == 10
            init = rand_vol(shape, 62)
            kept = init.copy()
            sv.affine(m, output=kept, keep_outside=True, _flags=_native.FORCE_TILED)
            got[(db, i)] = (out, kept)
        sv.close()
    for i, m in enumerate(mats):
        assert np.array_equal(got[('0', i)][0], got[('1', i)][0]), i
        assert np.array_equal(got[('0', i)][1], got[('1', i)][1]), i
        assert np.abs(got[('1', i)][0] - oracle.affine(vol, m, interp)).max() <= TOL[interp]

