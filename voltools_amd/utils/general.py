"""Host utilities: device strings and output-bounding-box arithmetic.

Mirrors the parts of ``/root/reference/voltools/utils/general.py`` that belong to the public API:

* ``get_available_devices`` (``general.py:61-80``): ``'cpu'`` always; ``'gpu'`` and ``'gpu:<i>'`` for every
  HIP device the native library can see.  The reference prints a warning when its GPU stack is
  missing (``:77-78``); so does this one.
* ``switch_to_device`` (``general.py:84-88``): the reference flips cupy's process-global current device;
  here the device index travels explicitly with every C-ABI call, so this only parses/validates.
* ``compute_post_transform_dimensions`` (``general.py:92-123``): padding needed by ``reshape=True``.

The reference's CUDA launch-geometry helpers (``general.py:9-58``) have no counterpart: tile shapes
and grids for gfx950 are chosen inside the native library per matrix.
"""
from typing import List, Tuple

import numpy as np

_warned = False


def parse_device(device: str) -> Tuple[str, int]:
    """``'cpu'`` -> ('cpu', -1); ``'gpu'`` -> ('gpu', 0); ``'gpu:3'`` -> ('gpu', 3)."""
    if device == 'cpu':
        return 'cpu', -1
    if device == 'gpu':
        return 'gpu', 0
    if device.startswith('gpu:') and device[4:].isdigit():
        return 'gpu', int(device[4:])
    raise ValueError(f'Unknown device ({device})')


def get_available_devices() -> List[str]:
    global _warned
    devices = ['cpu']
    count = 0
    try:
        from .. import _native
        count = _native.device_count()
    except OSError as e:       # library not built / HIP runtime missing
        if not _warned:
            print(f'Warning: the native HIP library could not be loaded ({e}). '
                  'Therefore, the only available device is "cpu".\n'
                  'Build it with: python -c "import __graft_entry__ as g; g.build()"')
            _warned = True
        return devices
    if count > 0:
        devices.append('gpu')
        devices.extend(f'gpu:{i}' for i in range(count))
    return devices


def switch_to_device(device: str) -> int:
    """Validate a device string and return the HIP device index (-1 for 'cpu')."""
    return parse_device(device)[1]


def compute_post_transform_dimensions(shape: Tuple[int, int, int], transform_m: np.ndarray) \
        -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Padding before/after and new dims so that the whole transformed box is kept (``reshape=True``).

    The 8 corners of the volume box ``[0, shape]`` are pushed through the inverse of the pull matrix,
    rounded to integers, and compared with the original extent (``general.py:92-123``).
    """
    extent = np.asarray(tuple(shape) + (1,), dtype=np.int64)
    corners = np.array([[(i >> a) & 1 for i in range(8)] for a in range(3)] + [[1] * 8]) * extent[:, None]
    try:
        inverse = np.linalg.inv(transform_m)
    except np.linalg.LinAlgError as e:
        print('Something went wrong. Transform matrix should have been affine but still couldnt inverse...')
        raise e
    moved = np.round(inverse @ corners).astype(int)

    pad_before = -np.minimum(moved, 0).min(axis=1)
    pad_after = np.maximum(moved - extent[:, None], 0).max(axis=1)
    new_dims = pad_before + extent + pad_after
    return pad_before[:3], pad_after[:3], new_dims[:3]
