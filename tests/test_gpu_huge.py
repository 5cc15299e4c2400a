"""-m gpu: one volume beyond 2^30 voxels -- 2048 x 1024 x 1024 float32 = 2^31 voxels, 8 GiB dense, 8.25 GiB resident.

SURVEY section 5 ("long-context" row) and section 7 (">2^32 voxels") name this regime; the reference cannot enter it at all (its kernel
indexes voxels with 32-bit `unsigned`/`int`, `/root/reference/voltools/transforms.py:243-262`, and a CUDA 3-D array of that depth
exceeds `cudaMalloc3DArray`'s 2048 limit on most parts).  Here the byte offsets of a launch that are kept in 32 bits (offsets inside a
buffer descriptor) are all relative to a per-tile / per-chunk base, so what has to hold is checked where it could break: linear
identity and integer shifts bit-exact on the device over all 2^31 voxels, blocks at the first, middle and LAST planes against the oracle
for the sweep (trilinear, cubic, prefiltered cubic), every general-matrix family against the direct kernel on the whole volume and
against the oracle on a crop in the far corner (largest offsets), and rotations about axes 1 / 2 (the exchanged resident copies).
The volume is generated on the device (torch: plumbing); only oracle windows cross PCIe.
"""
import ctypes

import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

pytestmark = pytest.mark.gpu

D, H, W = 2048, 1024, 1024
SHAPE = (D, H, W)


def centre():
    return np.divide(np.subtract(SHAPE, 1), 2, dtype=np.float32)


@pytest.fixture(scope='module')
def huge():
    torch = pytest.importorskip('torch')
    cu, lds, hbm = _native.device_props(0)
    if hbm < (96 << 30):
        pytest.skip('needs ~60 GiB of HBM')
    g = torch.Generator(device='cuda:0')
    g.manual_seed(2048)
    vol = torch.rand(SHAPE, dtype=torch.float32, device='cuda:0', generator=g)
    out = vt.empty(SHAPE, device='gpu:0')
    out2 = vt.empty(SHAPE, device='gpu:0')
    yield torch, vol, out, out2
    out.free()
    out2.free()
    del vol
    torch.cuda.empty_cache()
    _native.free_cached_memory(0)


def oracle_planes(vol, src_kind, m, d0, nb, ctx):
    a, b = max(0, d0 - ctx), min(D, d0 + nb + ctx)
    src = vol[a:b].cpu().numpy()
    if src_kind == 'filt':
        src = oracle.prefilter(src)
    kind = 'linear' if src_kind == 'linear' else 'bspline'
    return oracle.affine_ex(src, np.asarray(m, np.float64), kind, (nb, H, W), plane0=a, global_depth=D, out_plane0=d0)


def max_abs_diff(torch, a, b):
    # chunked: (a - b).abs() of 8 GiB tensors would allocate two more of them
    worst = 0.0
    for z in range(0, a.shape[0], 256):
        worst = max(worst, float((a[z:z + 256] - b[z:z + 256]).abs().max().item()))
    return worst


def test_huge_linear_identity_shift_sweep_and_general(huge):
    torch, vol, out, out2 = huge
    t_out = torch.as_tensor(out, device='cuda:0')
    t_out2 = torch.as_tensor(out2, device='cuda:0')
    sv = vt.StaticVolume(vol, interpolation='linear', device='gpu:0')
    info = sv.info()
    assert (info.depth, info.height, info.width) == SHAPE
    sv.affine(np.eye(4, dtype=np.float32), output=out)
    assert sv.info().last_kernel == 8
    sv.synchronize()
    assert bool(torch.equal(t_out, vol))
    sv.translate((7, -3, 11), output=out)
    sv.synchronize()
    assert bool(torch.equal(t_out[7:, :-3, 11:], vol[:-7, 3:, :-11]))
    assert float(t_out[:7].abs().max().item()) == 0.0 and float(t_out[:, -3:].abs().max().item()) == 0.0
    assert float(t_out[:, :, :11].abs().max().item()) == 0.0
    # the sweep: first, middle and last planes against the oracle
    for ang in (30.0, 100.0):
        m = vt.utils.transform_matrix(rotation=(0, ang, 0), rotation_units='deg', rotation_order='rzxz', center=centre())
        sv.affine(m, output=out)
        assert sv.info().last_kernel == 8
        sv.synchronize()
        for d0 in (0, 1020, D - 8):
            err = float(np.abs(out.get_planes(d0, d0 + 8) - oracle_planes(vol, 'linear', m, d0, 8, 2)).max())
            assert err <= 1e-6, (ang, d0, err)
    # general rotation (the reference's benchmark protocol): default dispatch and each general-matrix family vs the direct kernel
    m = vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=centre())
    sv.affine(m, output=out2, _flags=_native.FORCE_DIRECT)
    sv.synchronize()
    seen = set()
    for flags in (0, _native.NO_PACKED, _native.FORCE_PACKED):
        sv.affine(m, output=out, _flags=flags)
        seen.add(int(sv.info().last_kernel))
        sv.synchronize()
        assert max_abs_diff(torch, t_out, t_out2) <= 2e-6, flags
    assert seen <= {2, 6, 9} and len(seen) >= 2, seen
    # ... and the oracle on a crop in the far corner of the output (largest offsets on both sides)
    d0, h0, w0, e = D - 40, H - 36, W - 44, 24
    m64 = np.asarray(m, np.float64)
    corners = np.array([[d0 + a * e, h0 + b * e, w0 + c * e, 1.0] for a in (0, 1) for b in (0, 1) for c in (0, 1)])
    sc = corners @ m64[:3].T
    lo = np.maximum(np.floor(sc.min(0)).astype(int) - 2, 0)
    hi = np.minimum(np.ceil(sc.max(0)).astype(int) + 3, SHAPE)
    if np.all(hi > lo):
        sub = vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu().numpy()
        ms = m64.copy()
        ms[:3, 3] = m64[:3, :3] @ np.array([d0, h0, w0], np.float64) + m64[:3, 3] - lo
        want = oracle.affine_ex(sub, ms, 'linear', (e, e, e))
        got = out.get_planes(d0, d0 + e)[:, h0:h0 + e, w0:w0 + e]
        # (voxels whose source leaves the crop through a face that is a face of the volume are zero in both; faces cut inside the
        # volume are 2 voxels away from every tap)
        assert np.abs(got - want).max() <= 2e-6
    sv.close()


@pytest.mark.parametrize('interp', ['bspline', 'filt_bspline'])
def test_huge_cubic_sweep_and_axis_exchanges(interp, huge):
    torch, vol, out, out2 = huge
    t_out = torch.as_tensor(out, device='cuda:0')
    t_out2 = torch.as_tensor(out2, device='cuda:0')
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    filt = interp.startswith('filt_')
    tol = 3e-6 if filt else 1e-6
    if filt:
        assert float(sv.info().prefilter_ms) > 0
        sv.affine(np.eye(4, dtype=np.float32), output=out)
        sv.synchronize()
        worst = 0.0
        for z in range(16, D - 16, 254):
            z1 = min(z + 254, D - 16)
            worst = max(worst, float((t_out[z:z1, 16:-16, 16:-16] - vol[z:z1, 16:-16, 16:-16]).abs().max().item()))
        assert worst <= 5e-6, worst
    for ang in (30.0, 100.0):
        m = vt.utils.transform_matrix(rotation=(0, ang, 0), rotation_units='deg', rotation_order='rzxz', center=centre())
        sv.affine(m, output=out)
        assert sv.info().last_kernel == 8
        sv.synchronize()
        for d0 in (0, 1020, D - 8):
            want = oracle_planes(vol, 'filt' if filt else 'cubic', m, d0, 8, 40 if filt else 3)
            err = float(np.abs(out.get_planes(d0, d0 + 8) - want).max())
            assert err <= tol, (interp, ang, d0, err)
    # rotations about axis 1 march on an exchanged resident copy ([y][z][x]), rotations about axis 2 (integer offset) run the row kernel on the
    # plain one (kind 10) and, with VT_FORCE_XSWAP, march on the [x][y][z] copy; all against the general-matrix kernels
    for order_rot, want_kernel in (((0, 33, 0), 8), ((0, 0, 33), 10)):
        m = vt.utils.transform_matrix(rotation=order_rot, rotation_order='sxyz', center=centre())
        sv.affine(m, output=out)
        k = int(sv.info().last_kernel)
        sv.affine(m, output=out2, _flags=_native.NO_MARCH | _native.NO_ZSEP)
        k2 = int(sv.info().last_kernel)
        sv.synchronize()
        assert k == want_kernel and k2 in (2, 6, 9), (order_rot, k, k2)
        assert max_abs_diff(torch, t_out, t_out2) <= tol, (interp, order_rot)
        if want_kernel == 10:
            sv.affine(m, output=out, _flags=_native.FORCE_XSWAP)
            assert int(sv.info().last_kernel) == 8
            sv.synchronize()
            assert max_abs_diff(torch, t_out, t_out2) <= tol, (interp, order_rot, 'exchange path')
    sv.close()


def test_packed_spans_on_planes_beyond_2_31_bytes_per_box():
    """the packed-footprint kernels stage a box inside the volume through ONE buffer descriptor based at the box origin,
    so the 32-bit byte offset of the box's last row must stay below 2^31.  Planes of 4096 x 4128 floats (64.5 MiB) reach that at 32 box
    planes, which a general rotation whose w axis follows source axis 0 needs for a 16- or 8-deep tile (Lz = 33..45).  The planner must
    refuse such a box (pick_packed_tile, like plan_block) instead of letting the offset wrap: a wrapped offset lies beyond the descriptor's
    records, reads 0, and interior tiles come out silently wrong.  A thin volume with such planes, sampled on the plain copy
    (VT_NO_REORIENT: what the first three calls of a handle, slab handles and a failed reorientation do), forced to the packed family,
    against the direct kernel."""
    torch = pytest.importorskip('torch')
    cu, lds, hbm = _native.device_props(0)
    if hbm < (64 << 30):
        pytest.skip('needs ~20 GiB of HBM')
    shape = (40, 4096, 4096)
    g = torch.Generator(device='cuda:0')
    g.manual_seed(4096)
    vol = torch.rand(shape, dtype=torch.float32, device='cuda:0', generator=g)
    out = vt.empty(shape, device='gpu:0')
    out2 = vt.empty(shape, device='gpu:0')
    t_out = torch.as_tensor(out, device='cuda:0')
    t_out2 = torch.as_tensor(out2, device='cuda:0')
    c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    sv = vt.StaticVolume(vol, interpolation='linear', device='gpu:0')
    try:
        for rot in ((0.0, 78.0, 9.0), (5.0, 100.0, 0.0)):
            m = vt.utils.transform_matrix(rotation=rot, rotation_order='sxyz', center=c)
            sv.affine(m, output=out, _flags=_native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED | _native.NO_REORIENT)
            k = int(sv.info().last_kernel)
            lz = int(sv.info().last_lds_dims[0])
            if k == 6:
                assert lz * 4096 * 4128 * 4 < 2 ** 31, (rot, lz)                 # a packed box the descriptor can address
            sv.affine(m, output=out2, _flags=_native.FORCE_DIRECT)
            sv.synchronize()
            worst = 0.0
            for z in range(0, shape[0], 8):
                worst = max(worst, float((t_out[z:z + 8] - t_out2[z:z + 8]).abs().max().item()))
            assert worst <= 2e-6, (rot, k, lz, worst)
            assert float(t_out2.abs().max().item()) > 0.5                         # (the case does sample the volume)
    finally:
        sv.close()
        out.free()
        out2.free()
        del vol
        torch.cuda.empty_cache()
        _native.free_cached_memory(0)
