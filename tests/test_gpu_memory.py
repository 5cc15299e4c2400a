"""-m gpu: the resident-memory policy of a handle (round 5).

The reference keeps ONE CUDA array per StaticVolume (`/root/reference/voltools/volume.py:37-45`).  This library builds further resident
copies lazily, per orientation used (up to 13 buffers, 4x the volume for a sweep about one axis), so a handle has a
budget: `StaticVolume(..., max_resident_bytes=)` / `vt_volume_set_max_resident` / `VT_MAX_RESIDENT_GB`.  Held here:
  * results never depend on the budget (bit for bit against an unbudgeted handle, and against the oracle);
  * `info().resident_bytes` stays inside the budget after every call, copies are evicted least recently used first, and a copy that cannot
    fit at all is not built -- the call runs on the kernel family that samples the plain layout;
  * BASELINE config #4 (1024^3 `filt_bspline`, README sweep) runs inside 8.5 GiB -- the plain copy and ONE z-convolved plane-quad copy --
    on the marching kernel at every angle (the transposed orientation's quad form is built straight from the plain copy);
  * with HBM filled to within 3 GiB by another allocation (no lazy copy of a 1024^3 volume fits) the same sweep still agrees with the
    oracle, on the fallback families.
"""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = {'linear': 1e-6, 'bspline': 1e-6, 'filt_bspline': 3e-6}


def centre(shape):
    return np.divide(np.subtract(shape, 1), 2, dtype=np.float32)


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline'])
def test_budget_evicts_least_recently_used_and_changes_no_result(interp):
    shape = (96, 100, 104)
    vol = np.random.RandomState(5).random_sample(shape).astype(np.float32)
    c = centre(shape)
    free = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    plain = free.info().resident_bytes
    budget = int(plain * 2.6)                       # the plain copy, one more volume-sized copy and a bit: never two orientations at once
    tight = vt.StaticVolume(vol, interpolation=interp, device='gpu:0', max_resident_bytes=budget)
    assert tight.info().max_resident_bytes == budget
    rots = [(33, 0, 0), (0, 33, 0), (0, 0, 33), (80, 0, 0), (25, -40, 70), (0, 100, 0), (33, 0, 0), (0, 0, 120)]
    flags = _native.FORCE_TILED
    for i, rot in enumerate(rots):
        m = vt.utils.transform_matrix(rotation=rot, rotation_order='sxyz', center=c)
        a = free.affine(m, _flags=flags)
        b = tight.affine(m, _flags=flags)
        info = tight.info()
        assert info.resident_bytes <= budget, (interp, i, rot, info.resident_bytes, budget)
        want = oracle.affine(vol, m, interp)
        assert np.abs(b - want).max() <= TOL[interp], (interp, i, rot, info.last_kernel)
        if free.info().last_kernel == info.last_kernel:
            assert np.array_equal(a, b), (interp, i, rot, float(np.abs(a - b).max()))
    assert tight.info().copies_evicted > 0 and free.info().copies_evicted == 0
    assert free.info().resident_bytes > budget      # (the case does exercise the budget)
    assert free.info().copies_ms > 0.0 and free.info().copies_built >= 3
    # a budget below the present footprint releases copies at once; the plain copy alone always stays
    tight.set_max_resident(plain)
    assert tight.info().resident_bytes == plain
    m = vt.utils.transform_matrix(rotation=(33, 0, 0), rotation_order='sxyz', center=c)
    b = tight.affine(m, _flags=flags)
    assert tight.info().resident_bytes == plain and tight.info().last_kernel != 8      # no copy fits: a family on the plain layout
    assert np.abs(b - oracle.affine(vol, m, interp)).max() <= TOL[interp]
    tight.set_max_resident(0)
    tight.affine(m, _flags=flags)
    assert tight.info().last_kernel == 8 and tight.info().resident_bytes > plain
    free.close()
    tight.close()


N = 1024


@pytest.fixture(scope='module')
def big():
    torch = pytest.importorskip('torch')
    cu, lds, hbm = _native.device_props(0)
    if hbm < (64 << 30):
        pytest.skip('needs ~40 GiB of HBM')
    g = torch.Generator(device='cuda:0')
    g.manual_seed(1024)
    vol = torch.rand((N, N, N), dtype=torch.float32, device='cuda:0', generator=g)
    out = vt.empty((N, N, N), device='gpu:0')
    yield torch, vol, out
    out.free()
    del vol
    torch.cuda.empty_cache()
    _native.free_cached_memory(0)


def check_window(torch, vol, sv, out, angles, want_kernel=None, d0=500, nb=8, win=40):
    t_out = torch.as_tensor(out, device='cuda:0')
    window = vol[d0 - win:d0 + nb + win].cpu().numpy()
    src = oracle.prefilter(window)
    c = centre((N, N, N))
    kernels = []
    for ang in angles:
        m = vt.utils.transform_matrix(rotation=(0, ang, 0), rotation_units='deg', rotation_order='rzxz', center=c)
        sv.affine(m, output=out)
        sv.synchronize()
        kernels.append(int(sv.info().last_kernel))
        want = oracle.affine_ex(src, np.asarray(m, np.float64), 'bspline', (nb, N, N), plane0=d0 - win, global_depth=N, out_plane0=d0)
        got = t_out[d0:d0 + nb].cpu().numpy()
        assert np.abs(got - want).max() <= TOL['filt_bspline'], (ang, kernels[-1], float(np.abs(got - want).max()))
        if want_kernel is not None:
            assert kernels[-1] == want_kernel, (ang, kernels[-1])
    return kernels


def test_config4_sweep_inside_8p5_gib(big):
    torch, vol, out = big
    budget = int(8.5 * 2 ** 30)
    sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu:0', max_resident_bytes=budget)
    # the sweep crosses from the plain orientation to the in-plane transposed one at 45 degrees and back at 135
    for ang in (0.0, 30.0, 44.0):
        check_window(torch, vol, sv, out, [ang], want_kernel=8)
        assert sv.info().resident_bytes <= budget
    built_plain_side = sv.info().copies_built
    for ang in (46.0, 60.0, 90.0, 120.0, 134.0):
        check_window(torch, vol, sv, out, [ang], want_kernel=8)
        assert sv.info().resident_bytes <= budget, (ang, sv.info().resident_bytes)
    # ONE copy was built for the whole middle range -- the transposed orientation's z-convolved quad form, straight from the plain copy
    # (relayout_zquad_swap12; rounds 2-4 went through a volume-sized transposed plain copy) --, not one per angle
    assert sv.info().copies_built == built_plain_side + 1, (built_plain_side, sv.info().copies_built)
    for ang in (136.0, 170.0):
        check_window(torch, vol, sv, out, [ang], want_kernel=8)
        assert sv.info().resident_bytes <= budget
    assert sv.info().copies_built == built_plain_side + 2 and sv.info().copies_evicted >= 2
    sv.close()


def test_config4_parity_with_hbm_nearly_full(big):
    torch, vol, out = big
    sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu:0')
    _native.free_cached_memory(0)
    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info(0)
    leave = 3 << 30                                  # less than any lazy copy of this volume (4.1 GiB each)
    if free_b <= leave + (1 << 30):
        pytest.skip('not enough free HBM to stage the test')
    filler = []
    try:
        remaining = free_b - leave
        while remaining > (1 << 30):                 # in pieces: one allocation of ~270 GiB may exceed what a single hipMalloc serves
            piece = min(remaining, 64 << 30)
            filler.append(vt.empty((piece // 4,), device='gpu:0'))
            remaining -= piece
        kernels = check_window(torch, vol, sv, out, [0.0, 30.0, 100.0])
        assert all(k != 8 for k in kernels), kernels          # no plane-quad copy could be built: the families on the plain layout served
        assert sv.info().resident_bytes < (5 << 30)
    finally:
        for f in filler:
            f.free()
    # with the memory back the marching kernel returns (the retry throttle of a failed copy is cleared by release_copies)
    sv.release_copies()
    check_window(torch, vol, sv, out, [30.0], want_kernel=8)
    sv.close()


@pytest.mark.parametrize('interp', ['linear', 'filt_bspline', 'bspline_simple'])
@pytest.mark.parametrize('shape', [(41, 70, 133), (64, 64, 64), (9, 200, 37)])
def test_fused_transposing_relayout_builds_the_same_copy(interp, shape, monkeypatch):
    """The plane-quad forms of the in-plane transposed orientation (in-plane maps between 45 and 135 degrees) are built straight from the
    plain copy by `relayout_zquad_swap12` (round 5) instead of transpose02 + relayout_zquad[_fir] through an exchanged plain copy
    (`VT_NO_FUSED_RELAYOUT=1`): the same copy, so the marching kernel returns the same bits -- integer axis-0 offsets (the z-convolved copy
    of the cubic launches) and fractional ones (the plain plane-quad copy), ragged extents that leave partial 32 x 64 tiles and a partial
    last quad -- and no exchanged plain copy is left resident."""
    vol = np.random.RandomState(9).random_sample(shape).astype(np.float32)
    c = centre(shape)
    mats = []
    for ang, t0 in ((80.0, 0.0), (100.0, 2.0), (60.0, 0.5)):
        m = vt.utils.transform_matrix(rotation=(ang, 0, 0), rotation_order='sxyz', center=c)
        m[0, 3] += t0
        mats.append(m)
    res = {}
    for knob in ('0', '1'):
        monkeypatch.setenv('VT_NO_FUSED_RELAYOUT', knob)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        plain = sv.info().resident_bytes
        outs = []
        for m in mats:
            outs.append(sv.affine(m, _flags=_native.FORCE_TILED))
            assert sv.info().last_kernel == 8, (interp, shape, knob)
        res[knob] = (outs, sv.info().resident_bytes - plain, sv.info().copies_built)
        sv.close()
    for a, b in zip(res['0'][0], res['1'][0]):
        assert np.array_equal(a, b)
    assert res['0'][1] < res['1'][1] and res['0'][2] < res['1'][2]          # one volume-sized buffer and one build less
    tol = {'linear': 1e-6, 'filt_bspline': 3e-6, 'bspline_simple': 1e-6}[interp]
    for m, a in zip(mats, res['0'][0]):
        assert np.abs(a - oracle.affine(vol, m, interp)).max() <= tol
