"""Functional front ends + the ``affine`` dispatcher.

API mirror of ``/root/reference/voltools/transforms.py``: same function names, keyword arguments,
defaults, return/ownership rules and error types.

* ``device='cpu'`` -> ``scipy.ndimage.affine_transform`` with exactly the reference's arguments
  (``transforms.py:120-162``): ``order`` 1 for ``'linear'`` else 3, ``prefilter`` only for ``filt_bspline*``.
* ``device='gpu'|'gpu:N'`` -> the hand-written HIP library through the ctypes shim (``_native``):
  upload, optional three-pass prefilter, one transform kernel, download
  (the reference's GPU branch, ``transforms.py:164-226``).  There is no CPU fallback on this branch.

Differences from the reference, all deliberate (DESIGN.md "Semantics"):
  - coordinates are computed in float64 from the float32 matrix (the reference uses float32,
    ``transforms.py:269-274``), which removes an error that grows with the volume size;
  - outside voxels of a caller-supplied ``output=`` are written as 0 (the reference leaves them
    untouched, ``transforms.py:276-278``); pass ``keep_outside=True`` to ``StaticVolume.affine`` for the
    reference behaviour;
  - a device-resident *input* is never clobbered (the reference reuses it as the output buffer,
    ``transforms.py:196,208``).

One extension: the keyword-only ``edge=`` argument of the GPU devices.  ``'texture'`` (default) is the reference GPU path's
boundary contract (zero border, half-voxel skirt, ``transforms.py:187-191,276-278``); ``'scipy'`` is the contract of the
reference's CPU path (``scipy.ndimage.affine_transform(mode='constant', cval=0)``, ``transforms.py:147-152``: hard cut-off outside
``[0, dim-1]``, mirrored taps, mirror-boundary prefilter), so that ``device='gpu', edge='scipy'`` equals ``device='cpu'`` on the
WHOLE volume, not just where the two contracts agree -- the equivalence ``tests/test_devices.py:43-77`` of the reference eyeballs.
"""
import time
from typing import Tuple, Union

import numpy as np
from scipy.ndimage import affine_transform

from . import _native
from . import utils
from .utils import scale_matrix, shear_matrix, rotation_matrix, translation_matrix, transform_matrix

_INTERPOLATIONS = dict(_native.INTERP_CODES)        # same five names as transforms.py:11-17
AVAILABLE_INTERPOLATIONS = list(_INTERPOLATIONS.keys())
AVAILABLE_DEVICES = utils.get_available_devices()

Vec3 = Union[Tuple[float, float, float], np.ndarray]


def _triple(value):
    # the reference expands only python floats (transforms.py:42-45); ints fall through and fail later
    return (value, value, value) if isinstance(value, float) else value


def transform(volume: np.ndarray,
              scale: Union[float, Vec3] = None, shear: Union[float, Vec3] = None,
              rotation: Vec3 = None, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
              translation: Vec3 = None, center: Vec3 = None,
              interpolation: str = 'linear', reshape: bool = False, profile: bool = False,
              output=None, device: str = 'cpu', *, edge: str = 'texture'):
    """Scale, shear, rotate and translate about ``center`` (default ``(shape-1)/2``, transforms.py:38-39)."""
    if center is None:
        center = np.divide(np.subtract(volume.shape, 1), 2, dtype=np.float32)
    m = transform_matrix(_triple(scale), _triple(shear), rotation, rotation_units, rotation_order,
                         translation, center)
    return affine(volume, m, interpolation, reshape, profile, output, device, edge=edge)


def translate(volume: np.ndarray, translation: Vec3, interpolation: str = 'linear', reshape: bool = False,
              profile: bool = False, output=None, device: str = 'cpu', *, edge: str = 'texture'):
    return affine(volume, translation_matrix(translation), interpolation, reshape, profile, output, device, edge=edge)


def shear(volume: np.ndarray, coefficients: Union[float, Vec3], interpolation: str = 'linear',
          reshape: bool = False, profile: bool = False, output=None, device: str = 'cpu', *, edge: str = 'texture'):
    return affine(volume, shear_matrix(_triple(coefficients)), interpolation, reshape, profile, output, device, edge=edge)


def scale(volume: np.ndarray, coefficients: Union[float, Vec3], interpolation: str = 'linear',
          reshape: bool = False, profile: bool = False, output=None, device: str = 'cpu', *, edge: str = 'texture'):
    return affine(volume, scale_matrix(_triple(coefficients)), interpolation, reshape, profile, output, device, edge=edge)


def rotate(volume: np.ndarray, rotation: Vec3, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
           interpolation: str = 'linear', reshape: bool = False, profile: bool = False, output=None,
           device: str = 'cpu', *, edge: str = 'texture'):
    m = rotation_matrix(rotation=rotation, rotation_units=rotation_units, rotation_order=rotation_order)
    return affine(volume, m, interpolation, reshape, profile, output, device, edge=edge)


def _scipy_arguments(interpolation: str) -> Tuple[int, bool]:
    """order / prefilter selection of the CPU branch (transforms.py:126-134): anything that is not
    'linear' is cubic, only names starting with 'filt_bspline' prefilter."""
    order = 1 if interpolation == 'linear' else 3
    return order, interpolation.startswith('filt_bspline')


def _affine_cpu(volume, transform_m, interpolation, reshape, profile, output):
    t_start = time.time()
    order, prefilter = _scipy_arguments(interpolation)
    if reshape:
        pad_before, _, output_shape = utils.compute_post_transform_dimensions(volume.shape, transform_m)
        # scipy pads implicitly; shift the pull matrix by the leading pad (transforms.py:136-141)
        transform_m = np.dot(transform_m, translation_matrix(pad_before, transform_m.dtype))
    else:
        output_shape = volume.shape
    result = affine_transform(volume, transform_m, output_shape=output_shape, output=output,
                              order=order, prefilter=prefilter)
    if profile:
        print(f'transform finished in {(time.time() - t_start) * 1000:.3f}ms')
    return output if output is not None else result


def _affine_gpu(volume, transform_m, interpolation, reshape, profile, output, dev, edge='texture'):
    if interpolation not in _INTERPOLATIONS:
        # the reference raises from _get_transform_kernel (transforms.py:234-235)
        raise ValueError(f'Interpolation must be one of {AVAILABLE_INTERPOLATIONS}')
    lib = _native.load()
    t_wall = time.time()

    iface = getattr(volume, '__cuda_array_interface__', None)
    if iface is not None and not isinstance(volume, np.ndarray):
        # device-resident input: build a StaticVolume over it (not clobbered, unlike transforms.py:196,208)
        from .volume import StaticVolume
        if reshape:
            raise ValueError('reshape=True needs a host (numpy) volume')
        sv = StaticVolume(volume, interpolation=interpolation, device=f'gpu:{dev}', edge=edge)
        return sv.affine(transform_m, profile=profile, output=output)

    volume = np.asarray(volume)
    if volume.ndim != 3:
        raise ValueError('Expected a 3D array')
    if reshape:
        pad_before, pad_after, _ = utils.compute_post_transform_dimensions(volume.shape, transform_m)
        volume = np.pad(volume, list(zip(pad_before, pad_after)), mode='constant')
        # conjugate by the pad offset (transforms.py:171-178)
        transform_m = translation_matrix(-1 * pad_before) @ transform_m @ translation_matrix(pad_before)
    vol32 = np.ascontiguousarray(volume, dtype=np.float32)
    m32 = np.ascontiguousarray(np.asarray(transform_m, dtype=np.float32).reshape(4, 4))

    if output is None:
        host_out = _native.host_result(vol32.shape, dev)
        ptr, is_dev, fill = host_out.ctypes.data, False, None
    else:
        ptr, is_dev, fill = _native.resolve_output(output, vol32.shape, dev)
    if is_dev:
        # device output: resident path (upload + prefilter once, then the same kernel)
        from .volume import StaticVolume
        sv = StaticVolume(vol32, interpolation=interpolation, device=f'gpu:{dev}', edge=edge)
        return sv.affine(m32, profile=profile, output=output)

    import ctypes
    ms = ctypes.c_float(0.0)
    _native.check(lib.vt_affine_oneshot(dev, vol32.ctypes.data, *vol32.shape, _INTERPOLATIONS[interpolation],
                                        m32.ctypes.data, ptr, _native.ONESHOT_EDGE_SCIPY if edge == 'scipy' else 0,
                                        ctypes.byref(ms)), 'vt_affine_oneshot')
    if profile:
        print(f'transform finished in {ms.value:.3f}ms')
    return host_out if output is None else None      # GPU branch returns None when output= is given


EDGE_POLICIES = ('texture', 'scipy')


def affine(volume: np.ndarray, transform_m: np.ndarray, interpolation: str = 'linear', reshape: bool = False,
           profile: bool = False, output=None, device: str = 'cpu', *, edge: str = 'texture'):
    """Resample ``volume`` through the 4x4 pull matrix ``transform_m`` (transforms.py:109-229)."""
    if device not in AVAILABLE_DEVICES:
        raise ValueError(f'Unknown device ({device}), must be one of {AVAILABLE_DEVICES}')
    if edge not in EDGE_POLICIES:
        raise ValueError(f'edge must be one of {EDGE_POLICIES}')
    if device == 'cpu':
        return _affine_cpu(volume, transform_m, interpolation, reshape, profile, output)
    if device.startswith('gpu'):
        return _affine_gpu(volume, transform_m, interpolation, reshape, profile, output,
                           utils.switch_to_device(device), edge)
    raise ValueError(f'No instructions for {device}.')
