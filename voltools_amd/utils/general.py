"""Host utilities: device strings and output-bounding-box arithmetic.

Mirrors the parts of ``/root/reference/voltools/utils/general.py`` that belong to the public API:

* ``get_available_devices`` (``general.py:61-80``): ``'cpu'`` always; ``'gpu'`` and ``'gpu:<i>'`` for every
  HIP device the native library can see.  The reference prints a warning when its GPU stack is
  missing (``:77-78``); so does this one.
* ``switch_to_device`` (``general.py:84-88``): the reference flips cupy's process-global current device;
  here the device index travels explicitly with every C-ABI call, so this only parses/validates.
* ``compute_post_transform_dimensions`` (``general.py:92-123``): padding needed by ``reshape=True``.

* ``compute_prefilter_workgroup_dims`` / ``compute_elementwise_launch_dims`` (``general.py:9-58``): the
  reference's CUDA launch-geometry helpers, kept as public names.  They are INFORMATIONAL here: tile shapes
  and grids for gfx950 are chosen inside the native library per matrix, and nothing
  in this package launches with what they return.
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


def compute_prefilter_workgroup_dims(shape: Tuple[int, int, int]) -> Tuple[Tuple, Tuple]:
    """Informational: the thread-per-line launch geometry of the reference's three prefilter passes (``general.py:9-33``).

    Returns ``(grids, blocks)`` for the X, Y and Z pass like the reference: the block is the largest power-of-two
    rectangle that divides the two axes a pass runs over, at most 64 wide and 512 threads.  The native prefilter
    (``vt_kernels_prefilter.hip``) does not use it: it runs a fused X+Y pass and a block-form Z pass.
    """
    depth, height, width = (int(n) for n in shape)
    low_bit = lambda n: n & -n                  # largest power of two that divides n (0 for 0)
    bx = min(low_bit(width), low_bit(height), 64)
    by = min(low_bit(depth), low_bit(height), 512 // bx) if bx else 0
    if not bx or not by:
        raise ZeroDivisionError('empty volume')
    grids = ((height // bx, depth // by), (width // bx, depth // by), (width // bx, height // by))
    return grids, ((bx, by, 1),) * 3


def compute_elementwise_launch_dims(shape: Tuple[int, int, int]) -> Tuple[Tuple[int, int, int], Tuple[int, int, int]]:
    """Informational: the reference's grid-stride launch geometry (``general.py:36-58``) with gfx950's numbers.

    Wavefront size 64 in place of the warp size, the compute-unit count of GPU 0 (256 on an MI355X, also the
    value used when no GPU is visible) in place of the multiprocessor count.  The native transform kernels
    are tiled per matrix and never launched with this geometry.
    """
    wave, most = 64, 128
    cus = 256
    try:
        from .. import _native
        if _native.device_count() > 0:
            cus = int(_native.device_props(0)[0]) or cus
    except OSError:
        pass
    cap = 4 * 8 * cus                           # blocks: four rounds of eight resident blocks per compute unit
    n = int(np.prod(shape))
    waves = max(1, -(-n // wave))
    if waves <= cap:
        blocks, threads = waves, wave
    elif n < cap * most:
        blocks, threads = cap, -(-waves // cap) * wave
    else:
        blocks, threads = cap, most
    return (blocks, 1, 1), (threads, 1, 1)


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
