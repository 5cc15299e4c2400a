"""Slab-partitioned StaticVolume across GPUs (one process per GPU, torch.distributed: RCCL over xGMI on GPUs).

The reference is single-GPU (it only *selects* a device, ``/root/reference/voltools/utils/general.py:84-88``), so this
module has no counterpart there; it is BASELINE config #5 / SURVEY.md section 8(e).

Partition.  A global volume of ``G`` planes (axis 0) is cut into contiguous slabs, rank ``r`` owning planes
``[g0_r, g1_r)`` of both the source and the output.  Output voxels are independent, so a transform needs **no**
communication; the source needs one exchange, done once when the volume is built (the source is static):

* every rank keeps a *window* ``[g0 - h, g1 + h)`` of source planes resident, where the halo ``h`` is
  - the interpolation stencil: 1 plane (``linear``), 2 planes (cubic B-spline taps ``i-1 .. i+2``);
  - plus, for ``filt_*``, 16 planes of prefilter warm-up on each side: the recursive filter forgets its start
    after 16 samples (``|z|^16 = 7e-10``), so filtering the window reproduces the global filter on the planes
    that are actually sampled (overlap-and-discard; the reference's own initialisation horizon is 12,
    ``kernels/bspline.h:7``);
  - plus ``reach`` planes for transforms that move data along axis 0 (``reach=0`` covers every rotation about
    axis 0, i.e. the README sweep ``rotate((0, i, 0))``).  ``reach >= G`` degenerates into replicating the
    source on every rank (288 GB of HBM per GPU make that a valid fallback for general 3-D rotations).
* the planes a rank is missing are received point-to-point from whichever ranks own them (xGMI is
  point-to-point; neighbours only, unless ``h`` exceeds a slab) in one batch of ``isend/irecv``.

``SlabVolume.affine`` checks that the matrix' axis-0 reach fits the resident window and refuses otherwise
(no silent wrong answers).  Matrices are those of the *global* volume.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import ctypes
import math
import numpy as np

from . import _native
from .transforms import _INTERPOLATIONS
from .utils import switch_to_device, transform_matrix, rotation_matrix, translation_matrix

PREFILTER_WARMUP = 16


def stencil_halo(interpolation: str) -> int:
    if interpolation == 'linear':
        return 1
    return 2 + (PREFILTER_WARMUP if interpolation.startswith('filt_') else 0)


def slab_bounds(counts: Sequence[int]) -> List[Tuple[int, int]]:
    starts = np.concatenate([[0], np.cumsum(counts)])
    return [(int(starts[i]), int(starts[i + 1])) for i in range(len(counts))]


def plan_halo_exchange(counts: Sequence[int], rank: int, halo: int):
    """Who sends which global planes to whom.  Pure function (unit-tested without a process group).

    Returns ``(window, recvs, sends)``: ``window = (w0, w1)`` is the resident plane range of ``rank``;
    ``recvs = [(src_rank, a, b)]`` the plane ranges it receives; ``sends = [(dst_rank, a, b)]`` the ranges of its own
    planes it sends.  All ranges are global, half-open, clipped to ``[0, G)``.
    """
    bounds = slab_bounds(counts)
    total = bounds[-1][1]

    def window_of(r):
        g0, g1 = bounds[r]
        return max(0, g0 - halo), min(total, g1 + halo)

    def missing(r):
        w0, w1 = window_of(r)
        g0, g1 = bounds[r]
        return [(w0, g0), (g1, w1)]

    recvs, sends = [], []
    for other, (o0, o1) in enumerate(bounds):
        if other == rank:
            continue
        for a, b in missing(rank):                 # what I need from `other`
            lo, hi = max(a, o0), min(b, o1)
            if lo < hi:
                recvs.append((other, lo, hi))
        g0, g1 = bounds[rank]
        for a, b in missing(other):                # what `other` needs from me
            lo, hi = max(a, g0), min(b, g1)
            if lo < hi:
                sends.append((other, lo, hi))
    return window_of(rank), recvs, sends


def axis0_reach(matrix: np.ndarray, out_planes: Tuple[int, int], shape_hw: Tuple[int, int]) -> Tuple[float, float]:
    """Source-depth interval touched by output planes ``[d0, d1)`` x all (h, w) under row 0 of the pull matrix."""
    # separable min / max over the box's corners, in plain floats (this runs on the per-call path: ~1 us)
    m00, m01, m02, m03 = (float(x) for x in np.asarray(matrix).reshape(4, 4)[0])
    d0, d1 = out_planes
    H, W = shape_hw
    ed = (m00 * d0, m00 * (d1 - 1))
    eh = (0.0, m01 * (H - 1))
    ew = (0.0, m02 * (W - 1))
    return m03 + min(ed) + min(eh) + min(ew), m03 + max(ed) + max(eh) + max(ew)


class SlabVolume:
    """One rank's slab of a global volume, resident on its GPU; mirrors ``StaticVolume``'s transform methods.

    ``local`` holds this rank's planes (numpy array, or a torch tensor already on the target device).  ``group`` is a
    ``torch.distributed`` process group (``None`` = the default group).  ``engine`` replaces the HIP back end with a
    callable ``engine(window, plane0, global_depth, out_plane0, out_depth, interpolation)`` returning an object with
    ``affine(matrix, output)``; it exists so the exchange logic can be tested on CPU ranks over gloo.
    """

    def __init__(self, local, interpolation: str = 'linear', device: str = 'gpu', group=None, reach: int = 0,
                 engine: Optional[Callable] = None):
        import torch
        import torch.distributed as dist

        if interpolation not in _INTERPOLATIONS:
            raise ValueError(f'Interpolation must be one of {list(_INTERPOLATIONS)}')
        if local.ndim != 3:
            raise ValueError('Expected a 3D array')
        self.interpolation = interpolation
        self.device = device
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        on_gpu = device.startswith('gpu')
        self._dev = switch_to_device(device) if on_gpu else -1
        tdev = torch.device('cuda', self._dev) if on_gpu else torch.device('cpu')

        # The HIP path keeps a host slab on the host: its planes go straight into the resident buffer (one pinned upload), and only
        # the boundary planes a neighbour needs are staged as device tensors for the exchange.  The test engine (CPU ranks over
        # gloo) and device-resident inputs work on a tensor of the whole slab.
        host_slab = (engine is None) and not isinstance(local, torch.Tensor)
        if host_slab:
            own_np = np.ascontiguousarray(local, dtype=np.float32)
            own = None
            S, H, W = (int(s) for s in own_np.shape)
        else:
            own = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
            own = own.to(device=tdev, dtype=torch.float32).contiguous()
            S, H, W = (int(s) for s in own.shape)

        # slab sizes and cross-rank shape check
        meta = torch.tensor([S, H, W], dtype=torch.int64, device=tdev)
        metas = [torch.zeros_like(meta) for _ in range(self.world)]
        dist.all_gather(metas, meta, group=group)
        counts = [int(t[0]) for t in metas]
        if any(int(t[1]) != H or int(t[2]) != W for t in metas):
            raise ValueError('all slabs must share the in-plane shape')
        self.counts = counts
        self.bounds = slab_bounds(counts)
        self.global_shape = (sum(counts), H, W)
        self.shape = (S, H, W)                       # local output shape
        self.g0, self.g1 = self.bounds[self.rank]
        self.halo = stencil_halo(interpolation) + int(reach)

        (w0, w1), recvs, sends = plan_halo_exchange(counts, self.rank, self.halo)
        self.window = (w0, w1)

        def own_planes(a, b):                        # global planes [a, b) of this rank's slab as a device tensor
            if own is not None:
                return own[a - self.g0:b - self.g0].contiguous()
            return torch.from_numpy(own_np[a - self.g0:b - self.g0]).to(tdev)

        ops = [dist.P2POp(dist.isend, own_planes(a, b), self._global_rank(dst), group) for dst, a, b in sends]
        recv_bufs = []
        for src, a, b in recvs:
            buf = torch.empty((b - a, H, W), dtype=torch.float32, device=tdev)
            recv_bufs.append((buf, a, b))
            ops.append(dist.P2POp(dist.irecv, buf, self._global_rank(src), group))
        self.exchanged_bytes = sum((b - a) * H * W * 4 for _, a, b in recvs)
        self.sent_bytes = sum((b - a) * H * W * 4 for _, a, b in sends)
        # the path's only communication: timed (wall clock, this rank) so that bench.py can print it next to the step time
        import time
        if on_gpu:
            torch.cuda.synchronize(tdev)
        t_halo = time.perf_counter()
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if on_gpu:
            torch.cuda.synchronize(tdev)
        self.halo_ms = (time.perf_counter() - t_halo) * 1e3

        if engine is not None:
            window = torch.zeros((w1 - w0, H, W), dtype=torch.float32, device=tdev)
            window[self.g0 - w0:self.g1 - w0] = own
            for buf, a, b in recv_bufs:
                window[a - w0:b - w0] = buf
            self._engine = engine(window, w0, self.global_shape[0], self.g0, S, interpolation)
            self._handle = None
            del window
        else:
            if not on_gpu:
                raise ValueError("SlabVolume needs a GPU device (the HIP path has no CPU fallback)")
            self._engine = None
            self._lib = _native.load()
            # deferred handle: the resident window starts zero-filled; own planes and received halos are copied into it plane
            # range by plane range (no second buffer of the window's size), then the prefilter runs once over the whole window
            flags = _native.SRC_DEFERRED
            if w0 > 0:
                flags |= _native.SLAB_LO_INTERIOR
            if w1 < self.global_shape[0]:
                flags |= _native.SLAB_HI_INTERIOR
            h = ctypes.c_void_p()
            _native.check(self._lib.vt_volume_create_slab(self._dev, w1 - w0, H, W, _INTERPOLATIONS[interpolation], None, flags, w0,
                                                          self.global_shape[0], self.g0, S, ctypes.byref(h)),
                          'vt_volume_create_slab')
            self._handle = h
            if own is not None:
                _native.check(self._lib.vt_volume_upload_planes(h, self.g0 - w0, S, ctypes.c_void_p(own.data_ptr()), _native.SRC_DEVICE),
                              'vt_volume_upload_planes')
            else:
                _native.check(self._lib.vt_volume_upload_planes(h, self.g0 - w0, S, ctypes.c_void_p(own_np.ctypes.data), 0),
                              'vt_volume_upload_planes')
            for buf, a, b in recv_bufs:
                _native.check(self._lib.vt_volume_upload_planes(h, a - w0, b - a, ctypes.c_void_p(buf.data_ptr()), _native.SRC_DEVICE),
                              'vt_volume_upload_planes')
            _native.check(self._lib.vt_volume_finalize(h), 'vt_volume_finalize')
        del recv_bufs

    def _global_rank(self, group_rank: int) -> int:
        import torch.distributed as dist
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    # -- lifetime / introspection ------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, '_handle', None):
            self._lib.vt_volume_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> _native.VolumeInfo:
        info = _native.VolumeInfo()
        _native.check(self._lib.vt_volume_info(self._handle, ctypes.byref(info)), 'vt_volume_info')
        return info

    def synchronize(self) -> None:
        _native.check(self._lib.vt_volume_sync(self._handle), 'vt_volume_sync')

    def timer_start(self) -> None:
        _native.check(self._lib.vt_timer_start(self._handle), 'vt_timer_start')

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        _native.check(self._lib.vt_timer_stop(self._handle, ctypes.byref(ms)), 'vt_timer_stop')
        return ms.value

    # -- transforms -----------------------------------------------------------------------------------------------
    def check_reach(self, matrix: np.ndarray) -> None:
        """Refuse matrices whose axis-0 reach leaves the resident window (would silently read zeros)."""
        lo, hi = axis0_reach(matrix, (self.g0, self.g1), self.global_shape[1:])
        taps = 1 if self.interpolation == 'linear' else 2
        need_lo = max(0.0, math.floor(lo) - (taps - 1))
        need_hi = min(float(self.global_shape[0] - 1), math.floor(hi) + taps)
        warm = PREFILTER_WARMUP if self.interpolation.startswith('filt_') else 0
        have_lo = self.window[0] + (warm if self.window[0] > 0 else 0)
        have_hi = self.window[1] - 1 - (warm if self.window[1] < self.global_shape[0] else 0)
        if hi < -0.5 or lo >= self.global_shape[0] - 0.5:
            return                                     # the whole slab maps outside the volume: zeros are right
        if need_lo < have_lo or need_hi > have_hi:
            raise ValueError(f'rank {self.rank}: the transform reaches source planes [{need_lo:.0f}, {need_hi:.0f}] but planes '
                             f'[{have_lo}, {have_hi}] are resident; build the SlabVolume with a larger reach=')

    def affine(self, transform_m: np.ndarray, profile: bool = False, output=None) -> Optional[np.ndarray]:
        """Transform this rank's output planes ``[g0, g1)`` of the global volume (no communication)."""
        m = np.asarray(transform_m)
        self.check_reach(m)
        if self._engine is not None:
            return self._engine.affine(m, output)
        flags = 0
        if output is None:
            result = np.empty(self.shape, dtype=np.float32)
            ptr, is_dev = result.ctypes.data, False
        else:
            ptr, is_dev, _ = _native.resolve_output(output, self.shape, self._dev)
            result = None
        if is_dev:
            flags |= _native.OUT_DEVICE
        if profile:
            self.timer_start()
        if m.dtype == np.float64:
            m64 = np.ascontiguousarray(m.reshape(4, 4))
            rc = self._lib.vt_volume_affine_f64(self._handle, m64.ctypes.data, ptr, flags)
        else:
            m32 = np.ascontiguousarray(m, dtype=np.float32).reshape(4, 4)
            rc = self._lib.vt_volume_affine(self._handle, m32.ctypes.data, ptr, flags)
        _native.check(rc, 'vt_volume_affine')
        if profile:
            print(f'transform finished in {self.timer_stop():.3f}ms')
        return result

    def projection(self, transform_m: np.ndarray, reduce: bool = True, dst: Optional[int] = None):
        """``sum(axis=0)`` of the transformed *global* volume: each rank projects its own output planes
        (``vt_volume_project`` on the slab handle), then the (H, W) partials are summed over the group -- the path's
        one real exchange step: an RCCL all-reduce of H*W*4 bytes (1024^2: 4 MiB), or a reduce to group rank ``dst``.
        ``reduce=False`` returns this rank's partial sum.  Returns a float32 torch tensor (H, W) on the slab's device
        (the result on every rank, or on ``dst`` only)."""
        import torch
        import torch.distributed as dist
        m = np.asarray(transform_m)
        self.check_reach(m)
        H, W = self.global_shape[1:]
        if self._engine is not None:
            vol = self._engine.affine(m, None)
            part = torch.from_numpy(np.ascontiguousarray(vol.sum(axis=0, dtype=np.float64).astype(np.float32)))
        else:
            part = torch.empty((H, W), dtype=torch.float32, device=torch.device('cuda', self._dev))
            if m.dtype == np.float64:
                m64 = np.ascontiguousarray(m.reshape(4, 4))
                rc = self._lib.vt_volume_project_f64(self._handle, m64.ctypes.data, ctypes.c_void_p(part.data_ptr()),
                                                     _native.OUT_DEVICE)
            else:
                m32 = np.ascontiguousarray(m, dtype=np.float32).reshape(4, 4)
                rc = self._lib.vt_volume_project(self._handle, m32.ctypes.data, ctypes.c_void_p(part.data_ptr()),
                                                 _native.OUT_DEVICE)
            _native.check(rc, 'vt_volume_project')
            self.synchronize()                 # the handle's stream is not the stream RCCL enqueues on
        if reduce and self.world > 1:
            if dst is None:
                dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.reduce(part, dst=self._global_rank(dst), op=dist.ReduceOp.SUM, group=self.group)
        return part

    def transform(self, scale=None, shear=None, rotation=None, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
                  translation=None, center=None, profile: bool = False, output=None):
        if center is None:
            center = np.divide(np.subtract(self.global_shape, 1), 2, dtype=np.float32)
        if isinstance(scale, float):
            scale = (scale, scale, scale)
        if isinstance(shear, float):
            shear = (shear, shear, shear)
        m = transform_matrix(scale, shear, rotation, rotation_units, rotation_order, translation, center)
        return self.affine(m, profile, output)

    def rotate(self, rotation, rotation_units: str = 'deg', rotation_order: str = 'rzxz', profile: bool = False, output=None):
        return self.affine(rotation_matrix(rotation=rotation, rotation_units=rotation_units, rotation_order=rotation_order),
                           profile, output)

    def translate(self, translation, profile: bool = False, output=None):
        return self.affine(translation_matrix(translation), profile, output)
