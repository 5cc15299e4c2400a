"""4x4 pull-matrix builders (host side, numpy only).

Behavioural mirror of the reference's ``voltools/utils/matrices.py``:

* every matrix is a *pull* map in array-axis order (row 0 <-> axis 0), see
  ``/root/reference/voltools/transforms.py:147-152`` (scipy call) and ``:265-274`` (kernel);
* ``translation_matrix`` negates its argument (``matrices.py:22-27``);
* ``rotation_matrix`` negates the angles ("CCW notation", ``matrices.py:45``) and understands the
  24 Euler conventions ``s|r`` + three axis letters (``matrices.py:6-19``);
* ``transform_matrix`` composes ``T . C- . R . Sh . Sc . C+`` with float32 products, left to right
  (``matrices.py:111-154``) and normalises by ``m[3, 3]``.

The rotation is assembled here from elementary axis rotations instead of the closed-form Euler
table the reference uses; the two agree to float64 rounding before the final cast to ``dtype``
(pinned by ``tests/golden/matrices.npz``).
"""
from functools import reduce
from itertools import permutations
from typing import Sequence, Tuple, Union

import numpy as np

_AXIS_INDEX = {'x': 0, 'y': 1, 'z': 2}


def _euler_orders():
    """All 24 conventions: proper (aba) and Tait-Bryan (abc) sequences in both frames."""
    seqs = []
    for a, b, c in permutations('xyz', 3):
        seqs.append(a + b + c)
    for a, b in permutations('xyz', 2):
        seqs.append(a + b + a)
    return [f + s for f in 'sr' for s in sorted(seqs)]


AVAILABLE_ROTATIONS = _euler_orders()
AVAILABLE_UNITS = ['rad', 'deg']

Vec3 = Union[Tuple[float, float, float], Sequence[float], np.ndarray]


def _axis_rotation(axis: int, angle: float) -> np.ndarray:
    """Right-handed rotation by ``angle`` (radians) about coordinate axis ``axis`` (float64 3x3)."""
    c, s = np.cos(angle), np.sin(angle)
    a, b = (axis + 1) % 3, (axis + 2) % 3
    r = np.identity(3, dtype=np.float64)
    r[a, a] = c
    r[a, b] = -s
    r[b, a] = s
    r[b, b] = c
    return r


def translation_matrix(translation: Vec3, dtype=np.float32) -> np.ndarray:
    """Pull matrix of a shift by ``translation``: the offset column holds ``-translation``."""
    m = np.identity(4, dtype=dtype)
    m[:3, 3] = np.negative(np.asarray(translation[:3], dtype=dtype))
    return m


def rotation_matrix(rotation: Vec3, rotation_units: str = 'deg', rotation_order: str = 'rzxz',
                    dtype=np.float32) -> np.ndarray:
    """Pull matrix of an Euler rotation ``rotation=(a0, a1, a2)`` in convention ``rotation_order``.

    ``s...``: rotations about the fixed axes in the order written; ``r...``: about the rotating
    axes in the order written (equivalently fixed axes in reverse order).  Angles are negated
    first, exactly like the reference (``matrices.py:45``).
    """
    if rotation_units not in AVAILABLE_UNITS:
        raise ValueError(f'Rotation units must be one of {AVAILABLE_UNITS}')
    if rotation_order not in AVAILABLE_ROTATIONS:
        raise ValueError(f'Rotation order must be one of {AVAILABLE_ROTATIONS}')

    angles = np.asarray(rotation, dtype=np.float64)[:3]
    if rotation_units == 'deg':
        angles = np.deg2rad(angles)
    angles = -angles

    frame, axes = rotation_order[0], [_AXIS_INDEX[ch] for ch in rotation_order[1:]]
    steps = [_axis_rotation(ax, an) for ax, an in zip(axes, angles)]
    if frame == 's':
        steps.reverse()          # fixed axes: the first rotation is applied first -> rightmost
    r3 = reduce(np.matmul, steps)

    m = np.identity(4, dtype=dtype)
    m[:3, :3] = r3
    return m


def shear_matrix(coefficients: Vec3, dtype=np.float32) -> np.ndarray:
    """Upper-triangular shear: ``c[0]`` couples axis 0<-1, ``c[1]`` 0<-2, ``c[2]`` 1<-2."""
    m = np.identity(4, dtype=dtype)
    m[0, 1], m[0, 2], m[1, 2] = coefficients[0], coefficients[1], coefficients[2]
    return m


def scale_matrix(coefficients: Vec3, dtype=np.float32) -> np.ndarray:
    m = np.identity(4, dtype=dtype)
    m[0, 0], m[1, 1], m[2, 2] = coefficients[0], coefficients[1], coefficients[2]
    return m


def transform_matrix(scale: Vec3 = None, shear: Vec3 = None, rotation: Vec3 = None,
                     rotation_units: str = 'deg', rotation_order: str = 'rzxz',
                     translation: Vec3 = None, center: Vec3 = None, dtype=np.float32) -> np.ndarray:
    """Compose translation . (centre shift) . rotation . shear . scale . (centre shift back).

    Products are taken left to right in ``dtype`` so the float32 rounding is the reference's
    (``matrices.py:122-150``); the result is divided by ``m[3, 3]``.
    """
    factors = []
    if translation is not None:
        factors.append(translation_matrix(translation, dtype))
    if center is not None:
        factors.append(translation_matrix(tuple(-1 * c for c in center), dtype))
    if rotation is not None:
        factors.append(rotation_matrix(rotation, rotation_units, rotation_order, dtype))
    if shear is not None:
        factors.append(shear_matrix(shear, dtype))
    if scale is not None:
        factors.append(scale_matrix(scale, dtype))
    if center is not None:
        factors.append(translation_matrix(center, dtype))

    m = reduce(np.dot, factors, np.identity(4, dtype=dtype))
    m /= m[3, 3]
    return m
