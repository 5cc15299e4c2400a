"""-m gpu: BASELINE config #4 -- 1024^3 float32, StaticVolume reuse, README sweep (`/root/reference/README.md:25-27`,
`volume.py:61-91`) -- and the 1024^3 trilinear case the north star quotes its 70 % target on.

1024^3 takes planner branches no smaller volume reaches (a layer of tiles has 2048 workgroups, more than the chip keeps
resident: lookahead 2, no round-aware chunk count), so the kernels the sweep really runs are checked here: an 8-plane block of
the output against the CPU oracle (prefiltered on a +-40-plane window), the whole output against the other kernel families,
and bit-exact identity / integer shift for trilinear.  The volume is generated on the device (torch: plumbing) so that the
test does not move 4 GiB each way over PCIe; only the oracle's window comes back to the host.
"""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

pytestmark = pytest.mark.gpu

N = 1024
TOL = {'linear': 1e-6, 'filt_bspline': 3e-6}


def centre(shape):
    return np.divide(np.subtract(shape, 1), 2, dtype=np.float32)


@pytest.fixture(scope='module')
def big():
    torch = pytest.importorskip('torch')
    g = torch.Generator(device='cuda:0')
    g.manual_seed(1024)
    vol = torch.rand((N, N, N), dtype=torch.float32, device='cuda:0', generator=g)
    out = vt.empty((N, N, N), device='gpu:0')
    out2 = vt.empty((N, N, N), device='gpu:0')
    yield torch, vol, out, out2
    out.free()
    out2.free()
    del vol
    torch.cuda.empty_cache()
    _native.free_cached_memory(0)


@pytest.mark.parametrize('interp', ['filt_bspline', 'linear'])
def test_config4_1024_sweep_against_oracle_and_other_kernels(interp, big):
    torch, vol, out, out2 = big
    t_out = torch.as_tensor(out, device='cuda:0')
    t_out2 = torch.as_tensor(out2, device='cuda:0')
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    tol = TOL[interp]
    want_kernel = 8
    d0, nb, win = 500, 8, 40
    w0, w1 = d0 - win, d0 + nb + win
    window = vol[w0:w1].cpu().numpy()
    src = window if interp == 'linear' else oracle.prefilter(window)
    okind = 'linear' if interp == 'linear' else 'bspline'
    c = centre((N, N, N))
    for ang in (0.0, 30.0, 45.0, 100.0):
        # the README sweep's matrices: rotate((0, i, 0)) rzxz -- here about the centre, like bench.py (every voxel sampled)
        m = vt.utils.transform_matrix(rotation=(0, ang, 0), rotation_units='deg', rotation_order='rzxz', center=c)
        sv.affine(m, output=out)
        info = sv.info()
        assert info.last_kernel == want_kernel, (ang, info.last_kernel)
        got = out.get_planes(d0, d0 + nb)
        want = oracle.affine_ex(src, np.asarray(m, np.float64), okind, (nb, N, N), plane0=w0, global_depth=N, out_plane0=d0)
        err = float(np.abs(got - want).max())
        assert err <= tol, (interp, ang, err)
        # the other kernel families on the whole 1024^3 output (compared on the device)
        # (product build: the general-matrix kernels -- lane blocks / boxes / packed footprints -- on a separable matrix; test build
        # `make LEGACY=1`: round 1's marching kernels and the separable box kernel)
        legacy = _native.has_legacy_kernels()
        others = ((_native.NO_QUAD, (4,) if interp == 'linear' else (5,)), (_native.NO_ZPAIR, (4,)), (_native.NO_MARCH, (3,))) if legacy else \
                 ((_native.NO_QUAD, (2, 6, 9)), (_native.NO_QUAD | _native.NO_BLOCK, (2, 6)))
        for flags, k in others:
            if legacy and interp == 'linear' and flags == _native.NO_ZPAIR:
                continue
            sv.affine(m, output=out2, _flags=flags)
            assert sv.info().last_kernel in k, (ang, flags, sv.info().last_kernel)
            sv.synchronize()
            diff = float((t_out - t_out2).abs().max().item())
            assert diff <= tol, (interp, ang, flags, diff)
    if interp == 'linear':
        sv.affine(np.eye(4, dtype=np.float32), output=out)
        assert sv.info().last_kernel == 8
        sv.synchronize()
        assert bool(torch.equal(t_out, vol))
        sv.translate((7, -3, 11), output=out)
        sv.synchronize()
        assert bool(torch.equal(t_out[7:, :-3, 11:], vol[:-7, 3:, :-11]))
        assert float(t_out[:7].abs().max().item()) == 0.0 and float(t_out[:, -3:].abs().max().item()) == 0.0
    else:
        sv.affine(np.eye(4, dtype=np.float32), output=out)
        sv.synchronize()
        assert float((t_out - vol)[14:-14, 14:-14, 14:-14].abs().max().item()) <= 5e-6
    sv.close()


def test_config4_general_rotation_1024_sub_block(big):
    """The reference's own benchmark protocol uses general `sxyz` rotations (tests/benchmark.py:52-54): one of them at 1024^3,
    trilinear, default dispatch against the direct kernel on the whole volume and against the oracle on a small output block
    (the block's source window is cut out around its rotated bounding box)."""
    torch, vol, out, out2 = big
    t_out = torch.as_tensor(out, device='cuda:0')
    t_out2 = torch.as_tensor(out2, device='cuda:0')
    sv = vt.StaticVolume(vol, interpolation='linear', device='gpu:0')
    m = vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=centre((N, N, N)))
    sv.affine(m, output=out)
    k = sv.info().last_kernel
    assert k in (2, 6, 8, 9), k
    sv.affine(m, output=out2, _flags=_native.FORCE_DIRECT)
    sv.synchronize()
    assert float((t_out - t_out2).abs().max().item()) <= 2e-6
    # oracle on planes [508, 512): their source coordinates span all of axis 0, so use a cropped output block instead
    d0, h0, w0, e = 500, 480, 470, 24
    m64 = np.asarray(m, np.float64)
    corners = np.array([[d0 + a * e, h0 + b * e, w0 + c2 * e, 1.0] for a in (0, 1) for b in (0, 1) for c2 in (0, 1)])
    sc = corners @ m64[:3].T
    lo = np.maximum(np.floor(sc.min(0)).astype(int) - 2, 0)
    hi = np.minimum(np.ceil(sc.max(0)).astype(int) + 3, N)
    sub = vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu().numpy()
    # shift the problem into the crop: src_crop = M.(o + o0) - lo
    ms = m64.copy()
    ms[:3, 3] = m64[:3, :3] @ np.array([d0, h0, w0], np.float64) + m64[:3, 3] - lo
    want = oracle.affine_ex(sub, ms, 'linear', (e, e, e))
    got = out.get_planes(d0, d0 + e)[:, h0:h0 + e, w0:w0 + e]
    assert np.abs(got - want).max() <= 2e-6
    sv.close()
