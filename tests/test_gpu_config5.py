"""-m gpu: BASELINE config #5 at its OWN per-rank geometry, on one GPU.

Config #5 is an (8 x 1024) x 1024 x 1024 float32 volume in 8 axis-0 slabs of 1024^3, 'bspline', one slab per GPU.  The reference
has nothing like it (it only selects a device, `/root/reference/voltools/utils/general.py:84-88`, and its kernel indexes with
32-bit `unsigned`/`int`, `transforms.py:243-262`), so the only check of this path is ours.  What a rank of that job builds is
reproduced here exactly, through the same C-ABI calls `voltools_amd/distributed.py::SlabVolume` makes --
`vt_volume_create_slab(..., VT_SRC_DEFERRED, plane0 = g0 - halo, global_depth = 8192, out_plane0 = g0, out_depth = 1024)`,
`vt_volume_upload_planes` for the rank's own planes and for each halo it would have received, `vt_volume_finalize` -- for rank 7
(last: window [7166, 8192), 1026 planes) and rank 3 (interior: [3070, 4098), 1028 planes).  Planes are generated on the device
(torch: plumbing); the sweep runs about the GLOBAL centre; 8-plane output blocks at the slab's first planes (they tap the lower
halo), its middle and its last planes (upper halo / the global skirt) are compared with the oracle, and the whole 1024^3 slab
output with a whole-volume handle over the same window (a different host path: no slab offsets, matrix shifted by hand).
"""
import ctypes

import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from voltools_amd.distributed import SlabVolume, stencil_halo
from oracle import oracle

pytestmark = pytest.mark.gpu

WORLD, S, H, W = 8, 1024, 1024, 1024
G = WORLD * S
TOL = {'bspline': 1e-6, 'filt_bspline': 3e-6, 'linear': 1e-6}


def global_centre():
    return np.divide(np.subtract((G, H, W), 1), 2, dtype=np.float32)


def sweep_matrix(angle, tz=0.0):
    # the README sweep of the GLOBAL volume: rotate((0, i, 0)) rzxz about its centre (row 0 = [1 0 0 tz])
    return vt.utils.transform_matrix(rotation=(0, float(angle), 0), rotation_units='deg', rotation_order='rzxz',
                                     translation=(tz, 0, 0), center=global_centre())


class RankHandle:
    """What rank `rank` of the 8-rank job holds after SlabVolume.__init__, built call by call through the C ABI."""

    def __init__(self, torch, rank, interp, seed):
        self.lib = _native.load()
        self.interp = interp
        self.halo = stencil_halo(interp)
        self.g0, self.g1 = rank * S, (rank + 1) * S
        self.w0, self.w1 = max(0, self.g0 - self.halo), min(G, self.g1 + self.halo)
        gen = torch.Generator(device='cuda:0')
        gen.manual_seed(seed)
        self.window = torch.rand((self.w1 - self.w0, H, W), dtype=torch.float32, device='cuda:0', generator=gen)
        flags = _native.SRC_DEFERRED | (_native.SLAB_LO_INTERIOR if self.w0 > 0 else 0) | (_native.SLAB_HI_INTERIOR if self.w1 < G else 0)
        h = ctypes.c_void_p()
        _native.check(self.lib.vt_volume_create_slab(0, self.w1 - self.w0, H, W, _native.INTERP_CODES[interp], None, flags,
                                                     self.w0, G, self.g0, S, ctypes.byref(h)), 'vt_volume_create_slab')
        self.h = h
        # own planes, then each halo as its own upload (as received from the neighbour)
        for a, b in ((self.g0, self.g1), (self.w0, self.g0), (self.g1, self.w1)):
            if a < b:
                part = self.window[a - self.w0:b - self.w0]
                assert part.is_contiguous()
                _native.check(self.lib.vt_volume_upload_planes(h, a - self.w0, b - a, ctypes.c_void_p(part.data_ptr()), _native.SRC_DEVICE),
                              'vt_volume_upload_planes')
        _native.check(self.lib.vt_volume_finalize(h), 'vt_volume_finalize')

    def affine(self, m, out, flags=0):
        m64 = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(4, 4))
        _native.check(self.lib.vt_volume_affine_f64(self.h, m64.ctypes.data, ctypes.c_void_p(out.ptr), _native.OUT_DEVICE | flags),
                      'vt_volume_affine_f64')
        _native.check(self.lib.vt_volume_sync(self.h), 'vt_volume_sync')

    def info(self):
        info = _native.VolumeInfo()
        _native.check(self.lib.vt_volume_info(self.h, ctypes.byref(info)), 'vt_volume_info')
        return info

    def oracle_block(self, m, d0, nb, tz_planes=1):
        """Oracle on output planes [g0 + d0, g0 + d0 + nb) of the global volume, from the planes of the window those outputs can tap
        (filt_*: +-40 planes of context for the prefilter, clipped to the resident window)."""
        filt = self.interp.startswith('filt_')
        ctx = 40 if filt else 2 + tz_planes
        a = max(self.w0, self.g0 + d0 - ctx)
        b = min(self.w1, self.g0 + d0 + nb + ctx)
        src = self.window[a - self.w0:b - self.w0].cpu().numpy()
        if filt:
            src = oracle.prefilter(src)
        kind = {'linear': 'linear', 'bspline': 'bspline', 'filt_bspline': 'bspline'}[self.interp]
        return oracle.affine_ex(src, np.asarray(m, np.float64), kind, (nb, H, W), plane0=a, global_depth=G, out_plane0=self.g0 + d0)

    def close(self):
        if self.h:
            self.lib.vt_volume_destroy(self.h)
            self.h = None


def reach_checker(rank, interp):
    """SlabVolume.check_reach of the rank, without a process group (the method reads these attributes only)."""
    sv = object.__new__(SlabVolume)
    halo = stencil_halo(interp)
    sv.interpolation, sv.rank = interp, rank
    sv.g0, sv.g1 = rank * S, (rank + 1) * S
    sv.global_shape = (G, H, W)
    sv.window = (max(0, sv.g0 - halo), min(G, sv.g1 + halo))
    sv._handle = None
    return sv


@pytest.fixture(scope='module')
def outputs():
    torch = pytest.importorskip('torch')
    out = vt.empty((S, H, W), device='gpu:0')
    out2 = vt.empty((S, H, W), device='gpu:0')
    yield torch, out, out2
    out.free()
    out2.free()
    torch.cuda.empty_cache()
    _native.free_cached_memory(0)


@pytest.mark.parametrize('rank,interp', [(7, 'bspline'), (3, 'bspline'), (3, 'filt_bspline'), (0, 'linear')])
def test_config5_rank_handle_against_oracle(rank, interp, outputs):
    torch, out, out2 = outputs
    t_out = torch.as_tensor(out, device='cuda:0')
    t_out2 = torch.as_tensor(out2, device='cuda:0')
    rh = RankHandle(torch, rank, interp, seed=5000 + rank)
    assert (rh.w0, rh.w1) == {(7, 'bspline'): (7166, 8192), (3, 'bspline'): (3070, 4098), (3, 'filt_bspline'): (3054, 4114),
                              (0, 'linear'): (0, 1025)}[(rank, interp)]
    info = rh.info()
    assert (info.depth, info.out_depth) == (rh.w1 - rh.w0, S)
    chk = reach_checker(rank, interp)
    # the whole-volume twin: the same window as an ordinary volume of (w1 - w0) planes, output shape 1024^3, and the matrix moved by hand:
    # src_window = M . (d + g0, h, w, 1) - (w0, 0, 0)
    twin = vt.StaticVolume(rh.window, interpolation=interp, device='gpu:0')
    _native.check(rh.lib.vt_volume_set_output_shape(twin._handle, S, H, W), 'vt_volume_set_output_shape')
    tol = TOL[interp]
    blocks = (0, 508, S - 8)
    for ang, tz in ((0.0, 0.0), (30.0, 0.0), (100.0, 0.0), (45.0, 0.25)):
        m = sweep_matrix(ang, tz)
        assert tuple(np.asarray(m)[0, :3]) == (1.0, 0.0, 0.0)
        chk.check_reach(m)                                    # accepted: the sweep stays inside the stencil halo
        rh.affine(m, out)
        info = rh.info()
        assert info.last_kernel == 8, (ang, info.last_kernel)
        for d0 in blocks:
            got = out.get_planes(d0, d0 + 8)
            want = rh.oracle_block(m, d0, 8)
            err = float(np.abs(got - want).max())
            assert err <= tol, (rank, interp, ang, tz, d0, err)
            assert float(np.abs(want).max()) > 0.1            # the block is not vacuously zero
        m2 = np.asarray(m, np.float64).copy()
        m2[:3, 3] += m2[:3, 0] * rh.g0
        m2[0, 3] -= rh.w0
        m2 = np.ascontiguousarray(m2)
        _native.check(rh.lib.vt_volume_affine_f64(twin._handle, m2.ctypes.data, ctypes.c_void_p(out2.ptr), _native.OUT_DEVICE), 'vt_volume_affine_f64')
        twin.synchronize()
        # the twin treats the window's ends as the volume's: identical wherever no tap leaves the window and the skirt agrees, which is
        # every output plane here (the slab's outputs sit >= halo planes inside the window, or at the global end for the last rank)
        if interp.startswith('filt_'):
            # the twin's prefilter starts with the reference's boundary initialisation at the window's ends, the slab handle's with the
            # interior one: they differ by |z|^16 at the sampled planes
            assert float((t_out - t_out2).abs().max().item()) <= tol
        else:
            assert bool(torch.equal(t_out, t_out2)), (rank, interp, ang, float((t_out - t_out2).abs().max().item()))
    # refused: a general rotation, and an axis-0 shift beyond the halo (the planes it needs are on another GPU)
    with pytest.raises(ValueError):
        chk.check_reach(vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=global_centre()))
    with pytest.raises(ValueError):
        chk.check_reach(sweep_matrix(30.0, tz=-3.5 if rank == 0 else 3.5))     # translation t gives row 0 = [1 0 0 -t]
    twin.close()
    rh.close()


def test_config5_last_rank_skirt_and_outside_planes(outputs):
    """Rank 7's top planes under an axis-0 shift: output planes whose source falls beyond global plane 8191.5 - 0.5 are outside
    (zero), decided by the GLOBAL depth, not by the window's 1026 planes."""
    torch, out, _ = outputs
    rh = RankHandle(torch, 7, 'bspline', seed=77)
    m = sweep_matrix(30.0, tz=-0.75)                         # row 0 = [1 0 0 +0.75]: outputs d with d + 0.75 + 0.5 >= 8192 are outside: d = 8191
    assert float(np.asarray(m)[0, 3]) == 0.75
    rh.affine(m, out)
    got = out.get_planes(S - 8, S)
    want = rh.oracle_block(m, S - 8, 8)
    assert np.abs(got - want).max() <= 1e-6
    assert float(np.abs(got[-1]).max()) == 0.0 and float(np.abs(got[-2]).max()) > 0.1
    rh.close()
