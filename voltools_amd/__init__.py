"""MI355X-native 3-D affine resampling with the API surface of the-lay/voltools (v0.6.0).

``transform / affine / rotate / scale / shear / translate`` and ``StaticVolume`` keep the reference's
signatures (``/root/reference/voltools/__init__.py:3-5``); ``device='gpu'`` runs hand-written HIP kernels
for gfx950 through a C-ABI shared library, ``device='cpu'`` is scipy as in the reference.
"""
__version__ = '0.1.0'

from .transforms import AVAILABLE_INTERPOLATIONS, AVAILABLE_DEVICES, scale, shear, rotate, translate, transform, affine
from .volume import StaticVolume
from . import utils
from ._native import DeviceArray, free_cached_memory


def empty(shape, device: str = 'gpu') -> DeviceArray:
    """Uninitialised float32 device array usable as ``output=`` (stands in for ``cupy.empty``)."""
    return DeviceArray(shape, utils.switch_to_device(device))


def zeros(shape, device: str = 'gpu') -> DeviceArray:
    """Zero-filled float32 device array (stands in for ``cupy.zeros``, tests/benchmark.py:45)."""
    return DeviceArray(shape, utils.switch_to_device(device), zero=True)
