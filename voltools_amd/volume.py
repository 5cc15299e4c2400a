"""``StaticVolume``: upload (and prefilter) once, transform many times.

API mirror of ``/root/reference/voltools/volume.py:13-165``.  On a GPU device the constructor hands the
data to ``vt_volume_create`` (upload + one-time three-pass prefilter for ``filt_*``), and every
``affine`` is one 64-byte matrix hand-over plus one kernel launch (``volume.py:61-91``).  ``reshape`` is not
available here, as in the reference (``volume.py:14-16``).
"""
import ctypes
from typing import Tuple, Union

import numpy as np

from . import _native
from .transforms import affine as _affine, AVAILABLE_INTERPOLATIONS, _INTERPOLATIONS, _triple
from .utils import (scale_matrix, shear_matrix, rotation_matrix, translation_matrix, transform_matrix,
                    get_available_devices, switch_to_device)

Vec3 = Union[Tuple[float, float, float], np.ndarray]


class StaticVolume:
    """For StaticVolume transforms the boolean reshape cannot be given as an argument."""

    def __init__(self, data, interpolation: str = 'linear', device: str = 'gpu', *, edge: str = 'texture', max_resident_bytes: int = 0):
        """``edge`` (extension, GPU devices): ``'texture'`` = the reference GPU path's boundary contract, ``'scipy'`` = the
        contract of its CPU path (see ``transforms.py`` of this package).  ``max_resident_bytes`` (extension, GPU devices): how much HBM
        this volume may keep resident, its plain copy included -- the copies built lazily per orientation used (``info().resident_bytes``)
        are released least recently used first to stay inside it, and a copy that cannot fit is not built (the call then runs on the
        kernels that sample the plain layout; results are the same).  0 = no limit (or the ``VT_MAX_RESIDENT_GB`` environment default).
        The reference keeps one CUDA array per StaticVolume (``volume.py:37-45``)."""
        if data.ndim != 3:
            raise ValueError('Expected a 3D array')
        if edge not in ('texture', 'scipy'):
            raise ValueError("edge must be 'texture' or 'scipy'")
        self.edge = edge
        if device not in get_available_devices():
            raise ValueError(f'Unknown device ({device}), must be one of {get_available_devices()}')

        self.device = device
        self.interpolation = interpolation
        self.shape = tuple(int(s) for s in data.shape)
        self._handle = None

        if device.startswith('gpu'):
            if interpolation not in _INTERPOLATIONS:
                raise ValueError(f'Interpolation must be one of {AVAILABLE_INTERPOLATIONS}')
            self._dev = switch_to_device(device)
            self._lib = _native.load()
            iface = getattr(data, '__cuda_array_interface__', None)
            if iface is not None and not isinstance(data, np.ndarray):
                ptr, is_dev, _ = _native.resolve_output(data, self.shape, self._dev)
                flags = _native.SRC_DEVICE
            else:
                host = np.ascontiguousarray(data, dtype=np.float32)
                ptr, flags = host.ctypes.data, 0
            if edge == 'scipy':
                flags |= _native.EDGE_SCIPY
            h = ctypes.c_void_p()
            _native.check(self._lib.vt_volume_create(self._dev, *self.shape, _INTERPOLATIONS[interpolation],
                                                     ptr, flags, ctypes.byref(h)), 'vt_volume_create')
            self._handle = h
            self.d_type = np.dtype(np.float32)
            if max_resident_bytes:
                self.set_max_resident(int(max_resident_bytes))
        elif device == 'cpu':
            self.data = data

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, '_handle', None):
            self._lib.vt_volume_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection (no reference counterpart; used by bench.py and the tests) ------------------
    def info(self) -> _native.VolumeInfo:
        info = _native.VolumeInfo()
        _native.check(self._lib.vt_volume_info(self._handle, ctypes.byref(info)), 'vt_volume_info')
        return info

    def release_copies(self) -> int:
        """Free the resident copies this volume built lazily besides its plain one (exchanged orientations, plane-quad forms); they are
        rebuilt on demand.  Returns the bytes freed.  (No reference counterpart: the reference holds one CUDA array, volume.py:37-45.)"""
        if self.device == 'cpu':
            return 0
        freed = ctypes.c_uint64()
        _native.check(self._lib.vt_volume_release_copies(self._handle, ctypes.byref(freed)), 'vt_volume_release_copies')
        return int(freed.value)

    def set_max_resident(self, nbytes: int) -> None:
        """Change the resident-memory budget (bytes, the plain copy included; 0 = none).  See ``__init__``."""
        if self.device != 'cpu':
            _native.check(self._lib.vt_volume_set_max_resident(self._handle, ctypes.c_uint64(int(nbytes))), 'vt_volume_set_max_resident')

    def synchronize(self) -> None:
        _native.check(self._lib.vt_volume_sync(self._handle), 'vt_volume_sync')

    def timer_start(self) -> None:
        _native.check(self._lib.vt_timer_start(self._handle), 'vt_timer_start')

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        _native.check(self._lib.vt_timer_stop(self._handle, ctypes.byref(ms)), 'vt_timer_stop')
        return ms.value

    # -- the hot call -------------------------------------------------------------------------------
    def affine(self, transform_m: np.ndarray, profile: bool = False, output=None,
               keep_outside: bool = False, _flags: int = 0) -> Union[np.ndarray, None]:
        if self.device == 'cpu':
            return _affine(self.data, transform_m, interpolation=self.interpolation, profile=profile,
                           output=output, device=self.device)

        m = np.asarray(transform_m)
        flags = _flags | (_native.KEEP_OUTSIDE if keep_outside else 0)
        if output is None:
            result = _native.host_result(self.shape, self._dev)
            ptr, is_dev, fill = result.ctypes.data, False, None
            flags &= ~_native.KEEP_OUTSIDE        # a fresh buffer is zero outside (volume.py:73)
        else:
            ptr, is_dev, fill = _native.resolve_output(output, self.shape, self._dev)
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
        return result        # None when output= was given (volume.py:91)

    def transform(self, scale: Union[float, Vec3] = None, shear: Union[float, Vec3] = None,
                  rotation: Vec3 = None, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
                  translation: Vec3 = None, center: Vec3 = None, profile: bool = False,
                  output=None) -> Union[np.ndarray, None]:
        if center is None:
            center = np.divide(np.subtract(self.shape, 1), 2, dtype=np.float32)
        m = transform_matrix(_triple(scale), _triple(shear), rotation, rotation_units, rotation_order,
                             translation, center)
        return self.affine(m, profile, output)


    def affine_batch(self, matrices: np.ndarray, profile: bool = False, output=None,
                     _flags: int = 0) -> Union[np.ndarray, None]:
        """``affine`` for a stack of matrices (n, 4, 4) in one call.  Returns a float32 array
        (n, D, H, W), or fills ``output`` of that shape (numpy or device array) and returns None.  Small volumes
        (<= 96^3: template rotations) are served by a single kernel launch."""
        ms = np.ascontiguousarray(np.asarray(matrices, dtype=np.float32))
        if ms.ndim != 3 or ms.shape[1:] != (4, 4) or ms.shape[0] == 0:
            raise ValueError('matrices must have shape (n, 4, 4)')
        n = ms.shape[0]
        shape = (n,) + tuple(self.shape)
        if self.device == 'cpu':
            res = np.stack([_affine(self.data, m, interpolation=self.interpolation, device='cpu') for m in ms])
            if output is None:
                return res
            output[...] = res
            return output
        flags = _flags
        if output is None:
            result = _native.host_result(shape, self._dev)
            ptr, is_dev = result.ctypes.data, False
        else:
            ptr, is_dev, _ = _native.resolve_output(output, shape, self._dev)
            result = None
        if is_dev:
            flags |= _native.OUT_DEVICE
        if profile:
            self.timer_start()
        _native.check(self._lib.vt_volume_affine_batch(self._handle, n, ms.ctypes.data, ptr, flags), 'vt_volume_affine_batch')
        if profile:
            print(f'{n} transforms finished in {self.timer_stop():.3f}ms')
        return result

    # -- projection (examples/projections.py:20-26 does transform(...).sum(axis=0)) -------
    def projection(self, transform_m: np.ndarray, profile: bool = False, output=None,
                   _flags: int = 0) -> Union[np.ndarray, None]:
        """``affine(transform_m).sum(axis=0)`` without materialising the transformed volume.

        Returns a float32 (H, W) numpy array, or fills ``output`` (numpy / device array of that shape) and returns
        None, like ``affine``.  Rotations about axis 0 (the example's ``rotation=(i, 0, 0)``, ``'sxyz'``) cost one
        streaming pass over the resident volume plus a 2-D interpolation."""
        shape2 = tuple(self.shape[1:])
        if self.device == 'cpu':
            vol = _affine(self.data, transform_m, interpolation=self.interpolation, profile=profile, device='cpu')
            proj = vol.sum(axis=0, dtype=np.float64).astype(np.float32)
            if output is None:
                return proj
            output[...] = proj
            return output

        m = np.asarray(transform_m)
        flags = _flags
        if output is None:
            result = np.empty(shape2, dtype=np.float32)
            ptr, is_dev = result.ctypes.data, False
        else:
            ptr, is_dev, _ = _native.resolve_output(output, shape2, self._dev)
            result = None
        if is_dev:
            flags |= _native.OUT_DEVICE
        if profile:
            self.timer_start()
        if m.dtype == np.float64:
            m64 = np.ascontiguousarray(m.reshape(4, 4))
            rc = self._lib.vt_volume_project_f64(self._handle, m64.ctypes.data, ptr, flags)
        else:
            m32 = np.ascontiguousarray(m, dtype=np.float32).reshape(4, 4)
            rc = self._lib.vt_volume_project(self._handle, m32.ctypes.data, ptr, flags)
        _native.check(rc, 'vt_volume_project')
        if profile:
            print(f'projection finished in {self.timer_stop():.3f}ms')
        return result

    def project(self, scale: Union[float, Vec3] = None, shear: Union[float, Vec3] = None,
                rotation: Vec3 = None, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
                translation: Vec3 = None, center: Vec3 = None, profile: bool = False,
                output=None) -> Union[np.ndarray, None]:
        """``transform(...)`` followed by ``sum(axis=0)`` (same arguments as ``transform``)."""
        if center is None:
            center = np.divide(np.subtract(self.shape, 1), 2, dtype=np.float32)
        m = transform_matrix(_triple(scale), _triple(shear), rotation, rotation_units, rotation_order,
                             translation, center)
        return self.projection(m, profile, output)

    def translate(self, translation: Vec3, profile: bool = False, output=None):
        return self.affine(translation_matrix(translation), profile, output)

    def shear(self, coefficients: Union[float, Vec3], profile: bool = False, output=None):
        return self.affine(shear_matrix(_triple(coefficients)), profile, output)

    def scale(self, coefficients: Union[float, Vec3], profile: bool = False, output=None):
        return self.affine(scale_matrix(_triple(coefficients)), profile, output)

    def rotate(self, rotation: Vec3, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
               profile: bool = False, output=None):
        m = rotation_matrix(rotation=rotation, rotation_units=rotation_units, rotation_order=rotation_order)
        return self.affine(m, profile, output)
