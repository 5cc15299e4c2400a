"""ctypes binding of the C ABI declared in ``include/voltools_hip.h`` (``lib/libvoltools_hip.so``).

This is the thin shim the north star asks for: Python never touches HIP directly, it hands plain
pointers and sizes to the hand-written HIP library.  There is no CPU fallback behind these calls:
if the library is missing ``load()`` raises ``OSError``, and every non-zero return code becomes a
``RuntimeError`` carrying ``vt_last_error()``.
"""
import ctypes
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VT_LIB') or os.path.join(_HERE, 'lib', 'libvoltools_hip.so')   # VT_LIB: A/B experiments only

INTERP_CODES = {'linear': 0, 'bspline': 1, 'bspline_simple': 2, 'filt_bspline': 3, 'filt_bspline_simple': 4}

# enum vt_flags
OUT_DEVICE, KEEP_OUTSIDE, FORCE_DIRECT, FORCE_TILED, NO_ZSEP, NO_MARCH, NO_ZPAIR, NO_PACKED, FORCE_PACKED, FORCE_XSWAP, NO_RSWAP, NO_QUAD, ONESHOT_EDGE_SCIPY, NO_BLOCK, NO_ZFIR, NO_REORIENT, NO_ROWS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536
# enum vt_create_flags
SRC_DEVICE, SLAB_LO_INTERIOR, SLAB_HI_INTERIOR, SRC_DEFERRED, EDGE_SCIPY = 1, 2, 4, 8, 16

# every symbol include/voltools_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    'vt_device_count', 'vt_device_name', 'vt_device_props', 'vt_device_synchronize',
    'vt_malloc', 'vt_free', 'vt_memset_zero', 'vt_memcpy_h2d', 'vt_memcpy_d2h', 'vt_memcpy_d2d',
    'vt_host_register', 'vt_host_unregister', 'vt_device_trim',
    'vt_volume_create', 'vt_volume_create_slab', 'vt_volume_upload_planes', 'vt_volume_finalize', 'vt_volume_destroy', 'vt_volume_info', 'vt_volume_stream',
    'vt_volume_sync', 'vt_volume_set_output_shape', 'vt_volume_affine', 'vt_volume_affine_f64',
    'vt_volume_project', 'vt_volume_project_f64', 'vt_volume_affine_batch',
    'vt_timer_start', 'vt_timer_stop', 'vt_prefilter_inplace', 'vt_affine_oneshot',
    'vt_last_error', 'vt_version', 'vt_has_legacy_kernels', 'vt_volume_release_copies', 'vt_volume_set_max_resident',
]


class VolumeInfo(ctypes.Structure):
    _fields_ = [('device', ctypes.c_int32), ('interp', ctypes.c_int32),
                ('depth', ctypes.c_int32), ('height', ctypes.c_int32), ('width', ctypes.c_int32),
                ('out_depth', ctypes.c_int32), ('out_height', ctypes.c_int32), ('out_width', ctypes.c_int32),
                ('last_kernel', ctypes.c_int32), ('last_tile', ctypes.c_int32 * 3),
                ('last_lds_dims', ctypes.c_int32 * 3), ('last_lds_bytes', ctypes.c_int32),
                ('last_grid', ctypes.c_int32), ('prefilter_ms', ctypes.c_float),
                ('resident_bytes', ctypes.c_uint64), ('copies_ms', ctypes.c_float), ('copies_built', ctypes.c_int32),
                ('copies_evicted', ctypes.c_int32), ('max_resident_bytes', ctypes.c_uint64)]


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64;
    if this library pulled in the system copy first and torch was imported afterwards, the process would hold
    two HSA runtimes and torch would see no GPU.  When torch is installed (not necessarily imported) bind to
    its copy -- same SONAME, so our DT_NEEDED resolves to it.  VT_HIP_RUNTIME=system disables this."""
    import importlib.util
    import sys
    if os.environ.get('VT_HIP_RUNTIME', '') == 'system' or 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the library once; raises OSError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f'{LIB_PATH} not found (run __graft_entry__.build())')
    _preload_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_void_p, c_size_t, c_i64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64
    P = ctypes.POINTER
    L.vt_has_legacy_kernels.argtypes = []
    L.vt_device_count.argtypes = [P(c_int)]
    L.vt_device_name.argtypes = [c_int, ctypes.c_char_p, c_int]
    L.vt_device_props.argtypes = [c_int, P(c_int), P(c_int), P(ctypes.c_uint64)]
    L.vt_device_synchronize.argtypes = [c_int]
    L.vt_malloc.argtypes = [c_int, c_size_t, P(c_void_p)]
    L.vt_free.argtypes = [c_int, c_void_p]
    L.vt_memset_zero.argtypes = [c_int, c_void_p, c_size_t]
    L.vt_memcpy_h2d.argtypes = [c_int, c_void_p, c_void_p, c_size_t]
    L.vt_memcpy_d2h.argtypes = [c_int, c_void_p, c_void_p, c_size_t]
    L.vt_host_register.argtypes = [c_int, c_void_p, c_size_t]
    L.vt_host_unregister.argtypes = [c_int, c_void_p]
    L.vt_device_trim.argtypes = [c_int]
    L.vt_memcpy_d2d.argtypes = [c_int, c_void_p, c_void_p, c_size_t]
    L.vt_volume_create.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, P(c_void_p)]
    L.vt_volume_create_slab.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                        c_i64, c_i64, c_i64, c_int, P(c_void_p)]
    L.vt_volume_upload_planes.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int]
    L.vt_volume_finalize.argtypes = [c_void_p]
    L.vt_volume_destroy.argtypes = [c_void_p]
    L.vt_volume_info.argtypes = [c_void_p, P(VolumeInfo)]
    L.vt_volume_stream.argtypes = [c_void_p, P(c_void_p)]
    L.vt_volume_sync.argtypes = [c_void_p]
    L.vt_volume_release_copies.argtypes = [c_void_p, P(ctypes.c_uint64)]
    L.vt_volume_set_max_resident.argtypes = [c_void_p, ctypes.c_uint64]
    L.vt_volume_set_output_shape.argtypes = [c_void_p, c_int, c_int, c_int]
    L.vt_volume_affine.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    L.vt_volume_affine_f64.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    L.vt_volume_project.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    L.vt_volume_affine_batch.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int]
    L.vt_volume_project_f64.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    L.vt_timer_start.argtypes = [c_void_p]
    L.vt_timer_stop.argtypes = [c_void_p, P(ctypes.c_float)]
    L.vt_prefilter_inplace.argtypes = [c_int, c_void_p, c_int, c_int, c_int]
    L.vt_affine_oneshot.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                    P(ctypes.c_float)]
    L.vt_last_error.restype = ctypes.c_char_p
    L.vt_version.restype = ctypes.c_char_p
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ('vt_last_error', 'vt_version'):
            fn.restype = c_int
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vt_last_error()
        raise RuntimeError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


def has_legacy_kernels() -> bool:
    """True for the test build (`make LEGACY=1`: round 1's marching kernels 4 / 5 and the axis-0-separable box kernel 3 compiled in)."""
    return bool(load().vt_has_legacy_kernels())


def device_count() -> int:
    n = ctypes.c_int(0)
    check(load().vt_device_count(ctypes.byref(n)), 'vt_device_count')
    return n.value


def free_cached_memory(dev: int = 0) -> None:
    """Release the device buffers the library keeps for recycling (at most 16 GiB per device) and the pinned host result
    buffers of the pool; the reference's users call ``cp.get_default_memory_pool().free_all_blocks()`` for the same purpose."""
    check(load().vt_device_trim(dev), 'vt_device_trim')
    _host_pool.clear(dev)


def device_name(dev: int) -> str:
    buf = ctypes.create_string_buffer(256)
    check(load().vt_device_name(dev, buf, 256), 'vt_device_name')
    return buf.value.decode()


def device_props(dev: int) -> Tuple[int, int, int]:
    cu, lds, hbm = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
    check(load().vt_device_props(dev, ctypes.byref(cu), ctypes.byref(lds), ctypes.byref(hbm)), 'vt_device_props')
    return cu.value, lds.value, hbm.value


class DeviceArray:
    """A float32 array in HBM owned through vt_malloc/vt_free (what ``cp.zeros``/``cp.asarray`` were to
    the reference, transforms.py:180, volume.py:73).  Exposes ``__cuda_array_interface__`` so torch-ROCm can
    wrap it without a copy."""

    def __init__(self, shape, device: int = 0, zero: bool = False):
        self.shape = tuple(int(s) for s in shape)
        self.ndim = len(self.shape)
        self.dtype = np.dtype(np.float32)
        self.device = int(device)
        self.size = int(np.prod(self.shape))
        self.nbytes = self.size * 4
        p = ctypes.c_void_p()
        check(load().vt_malloc(self.device, max(self.nbytes, 4), ctypes.byref(p)), 'vt_malloc')
        self.ptr = p.value
        if zero:
            self.fill_zero()

    @classmethod
    def from_numpy(cls, a: np.ndarray, device: int = 0) -> 'DeviceArray':
        a = np.ascontiguousarray(a, dtype=np.float32)
        d = cls(a.shape, device)
        check(load().vt_memcpy_h2d(device, d.ptr, a.ctypes.data, a.nbytes), 'vt_memcpy_h2d')
        return d

    def fill_zero(self) -> None:
        check(load().vt_memset_zero(self.device, self.ptr, self.nbytes), 'vt_memset_zero')

    def get(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        check(load().vt_memcpy_d2h(self.device, out.ctypes.data, self.ptr, self.nbytes), 'vt_memcpy_d2h')
        return out

    def get_planes(self, d0: int, d1: int) -> np.ndarray:
        """Planes [d0, d1) along axis 0 (a 1024^3 result is 4 GiB: tests and tools look at a few planes of it)."""
        d0, d1 = int(d0), int(d1)
        if not (0 <= d0 < d1 <= self.shape[0]):
            raise ValueError(f'plane range [{d0}, {d1}) outside 0..{self.shape[0]}')
        plane = int(np.prod(self.shape[1:])) * 4
        out = np.empty((d1 - d0,) + self.shape[1:], dtype=np.float32)
        check(load().vt_memcpy_d2h(self.device, out.ctypes.data, ctypes.c_void_p(self.ptr + d0 * plane), out.nbytes), 'vt_memcpy_d2h')
        return out

    def set(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != self.shape:
            raise ValueError(f'shape mismatch {a.shape} vs {self.shape}')
        check(load().vt_memcpy_h2d(self.device, self.ptr, a.ctypes.data, a.nbytes), 'vt_memcpy_h2d')

    @property
    def __cuda_array_interface__(self):
        return {'shape': self.shape, 'typestr': '<f4', 'data': (self.ptr, False), 'version': 2, 'strides': None}

    def free(self) -> None:
        if getattr(self, 'ptr', None):
            load().vt_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _HostResultPool:
    """Result arrays for calls without ``output=``.

    The reference returns ``cupy_array.get()`` (transforms.py:223, volume.py:89): a freshly allocated numpy array.
    Writing 512 MiB into fresh pages costs ~45 ms of page faults on the GPU box -- 5x the PCIe transfer itself
    (tools/pcie_probe.py).  Results therefore come from a small pool of host buffers that stay faulted-in and
    registered with the HIP runtime: a buffer is handed out again once the caller has dropped every array that
    views it (reference count of the backing array back at the pool's own).  Arrays the caller keeps are never reused.
    Every pooled buffer is an anonymous mapping of its own, 2 MiB-aligned, whole 2 MiB units (``_page_aligned_backing``).
    ``VT_HOST_POOL=0`` disables the pool; ``VT_HOST_POOL_MB`` caps it (default: a tenth of the machine's memory, between
    4 and 64 GiB -- a loop over 1024^3 one-shot results alternates between two 4-GiB buffers: 94 ms per call with both in the
    pool, 250 ms when one of them is a fresh allocation every time)."""

    def __init__(self):
        self.entries = []          # (backing uint8 array, offset of the first page boundary, floats, registered bytes); most recently used last
        self.enabled = os.environ.get('VT_HOST_POOL', '1') != '0'
        try:
            phys_mb = (os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES')) >> 20
        except (ValueError, OSError, AttributeError):
            phys_mb = 0
        default_mb = min(65536, max(4096, phys_mb // 10))
        self.cap = int(os.environ.get('VT_HOST_POOL_MB', str(default_mb))) << 20
        # Results below 1 MiB are not worth a pool slot (a 2 MiB-granular one): they are plain numpy arrays, copied through the runtime's
        # staging buffers.  From 1 MiB on a result that is NOT pooled would travel in 512 KiB staged slices (the library never lets the runtime
        # pin caller memory in place, and never registers anything the process heap could hold: DESIGN.md section 6) -- 100^3: 0.34 ms instead
        # of 0.12 --, so the pool starts there (round 4 had it at 8 MiB, the then threshold of the library's own in-place pinning).
        self.min_bytes = int(float(os.environ.get('VT_HOST_POOL_MIN_MB', '1')) * (1 << 20))
        # reference count of a backing array that nobody outside the pool refers to, measured by the scan itself
        # (the entry is built inside the call: a local name for the backing array would count as a holder)
        self._idle = self._scan([self._new_entry(1)], 1, calibrate=True)

    def _scan(self, entries, n, calibrate=False) -> int:
        """Index of a free buffer of n floats (most recently used first), or -1.  Every array handed out is a view whose
        base is the backing array (numpy collapses view chains), so its reference count tells whether a caller still holds one."""
        import sys
        for i in range(len(entries) - 1, -1, -1):
            if entries[i][2] == n:
                c = sys.getrefcount(entries[i][0])
                if calibrate:
                    return c
                if c <= self._idle:
                    return i
        return -1

    @staticmethod
    def _new_entry(n: int):
        raw, off, reg = _page_aligned_backing(n)
        return raw, off, n, reg

    @staticmethod
    def _view(entry, shape) -> np.ndarray:
        raw, off, n, _ = entry
        return raw[off:off + 4 * n].view(np.float32).reshape(shape)

    @staticmethod
    def _addr(entry) -> int:
        return entry[0].ctypes.data + entry[1]

    def take(self, shape, device: int) -> np.ndarray:
        n = int(np.prod(shape))
        if not self.enabled or n * 4 < self.min_bytes:
            return np.empty(shape, dtype=np.float32)
        i = self._scan(self.entries, n)
        if i >= 0:
            self.entries.append(self.entries.pop(i))
            return self._view(self.entries[-1], shape)
        # make room: drop free buffers of other sizes, oldest first
        total = sum(e[3] for e in self.entries) + n * 4
        while total > self.cap:
            j = -1
            for size in sorted({e[2] for e in self.entries}):
                j = self._scan(self.entries[::-1], size)        # reversed: oldest first
                if j >= 0:
                    j = len(self.entries) - 1 - j
                    break
            if j < 0:
                break
            victim = self.entries.pop(j)
            self._unregister(device, victim)
            total -= victim[3]
            del victim
        entry = self._new_entry(n)
        if total <= self.cap and load().vt_host_register(device, ctypes.c_void_p(self._addr(entry)), entry[3]) == 0:
            self.entries.append(entry)
        return self._view(entry, shape)

    def _unregister(self, device: int, entry) -> None:
        rc = load().vt_host_unregister(device, ctypes.c_void_p(self._addr(entry)))
        if rc != 0:                # someone else removed the registration: a bug worth hearing about, not worth failing a result for
            import warnings
            warnings.warn(f'host result pool: unregister of a {entry[3]}-byte buffer failed (code {rc})', RuntimeWarning)

    def clear(self, device: int = 0) -> None:
        for e in self.entries:
            self._unregister(device, e)
        self.entries = []


_PAGE = 4096
_HUGE = 2 << 20          # granularity at which the pool isolates its buffers: the host's transparent huge pages


def _page_aligned_backing(n: int):
    """Backing store for a float32 result of n elements: (uint8 array, offset of the registered range in it, bytes to register).
    Registration pins and maps pages at the host's own virtual address, and releasing ANY pinned range that shares a mapping unit with
    another takes the shared unit out of the GPU's page table under the other: a `Memory access fault by GPU` on a page-aligned
    HOST address at some later copy (the round-1 abort inside vt_volume_create; round 2, with the runtime's message intact).  The unit
    is not always a 4 KiB page: heap arrays of a few MiB sit next to each other inside 2 MiB transparent huge pages, and the runtime's
    own temporary pin of a pageable transfer (the upload of the next volume, say) covers whole units (round 4: the fault again, one
    GPU box in three, on a 4.8 MiB result that shared a huge page with the test's input array).  A pooled buffer therefore owns an
    anonymous mapping of its own, 2 MiB-aligned and a whole number of 2 MiB units: nothing else can live in a unit that gets registered."""
    import mmap
    reg = (n * 4 + _HUGE - 1) // _HUGE * _HUGE
    raw = np.frombuffer(mmap.mmap(-1, reg + _HUGE), dtype=np.uint8)
    return raw, (-raw.ctypes.data) % _HUGE, reg


_host_pool = _HostResultPool()


def host_result(shape, device: int = 0) -> np.ndarray:
    """A float32 host array for a result that is returned to the caller (see _HostResultPool)."""
    return _host_pool.take(tuple(int(s) for s in shape), device)


def resolve_output(output, shape, device: int) -> Tuple[Optional[int], bool, Optional[np.ndarray]]:
    """Classify an ``output=`` argument -> (pointer, is_device, numpy array to fill or None).

    Accepted: ``DeviceArray``; anything exposing ``__cuda_array_interface__`` (torch-ROCm tensors,
    cupy-on-ROCm arrays) -- float32, C-contiguous, right shape; or a numpy float32 array (host).
    """
    shape = tuple(int(s) for s in shape)
    if isinstance(output, np.ndarray):
        if output.dtype != np.float32 or not output.flags.c_contiguous or output.shape != shape:
            raise ValueError('output must be a C-contiguous float32 array of the volume shape')
        return output.ctypes.data, False, output
    iface = getattr(output, '__cuda_array_interface__', None)
    if iface is None:
        raise TypeError(f'unsupported output type {type(output)}')
    if tuple(iface['shape']) != shape or iface['typestr'] not in ('<f4', '=f4', 'f4'):
        raise ValueError('device output must be float32 of the volume shape')
    if iface.get('strides') is not None:
        expect = tuple(int(np.prod(shape[i + 1:])) * 4 for i in range(len(shape)))
        if tuple(iface['strides']) != expect:
            raise ValueError('device output must be C-contiguous')
    dev_attr = getattr(output, 'device', None)
    dev_index = getattr(dev_attr, 'index', dev_attr if isinstance(dev_attr, int) else None)
    if dev_index is not None and dev_index != device:
        raise ValueError(f'output lives on device {dev_index}, volume on device {device}')
    return int(iface['data'][0]), True, None
