"""ctypes loader for the CPU parity oracle (oracle/vt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by anything under voltools_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libvt_oracle.so')

FAITHFUL = 1
KEEP_OUTSIDE = 2
TEXFRAC8 = 4

INTERP = {'linear': 0, 'bspline': 1, 'bspline_simple': 2, 'filt_bspline': 3, 'filt_bspline_simple': 4}

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'vt_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B' if force else '-s'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
        f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
        i64 = ctypes.c_int64
        L.vt_oracle_affine.argtypes = [f32p, i64, i64, i64, f32p, f32p, ctypes.c_int, ctypes.c_int]
        L.vt_oracle_affine.restype = ctypes.c_int
        L.vt_oracle_affine_ex.argtypes = [f32p, i64, i64, i64, i64, i64, f32p, i64, i64, i64, i64, f64p,
                                          ctypes.c_int, ctypes.c_int]
        L.vt_oracle_affine_ex.restype = ctypes.c_int
        L.vt_oracle_prefilter.argtypes = [f32p, i64, i64, i64]
        L.vt_oracle_prefilter.restype = None
        L.vt_oracle_prefilter_line.argtypes = [f32p, ctypes.c_uint32, ctypes.c_ssize_t]
        L.vt_oracle_prefilter_line.restype = None
        L.vt_oracle_num_threads.restype = ctypes.c_int
        L.vt_oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _interp_code(interpolation):
    if isinstance(interpolation, str):
        return INTERP[interpolation]
    return int(interpolation)


def prefilter(volume: np.ndarray) -> np.ndarray:
    """Three-pass B-spline prefilter (bspline.h:30-99) on a copy."""
    v = np.ascontiguousarray(volume, dtype=np.float32).copy()
    lib().vt_oracle_prefilter(v, *v.shape)
    return v


def prefilter_line(line: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(line, dtype=np.float32).copy()
    lib().vt_oracle_prefilter_line(v, v.size, 1)
    return v


def affine(volume: np.ndarray, m: np.ndarray, interpolation='linear', flags: int = 0,
           output: np.ndarray = None, prefiltered: bool = False) -> np.ndarray:
    """Reference GPU-path semantics of ``affine`` (transforms.py:164-226) on the CPU.

    ``filt_*`` interpolations prefilter a private copy first unless ``prefiltered`` is set.
    """
    code = _interp_code(interpolation)
    v = np.ascontiguousarray(volume, dtype=np.float32)
    if code >= 3 and not prefiltered:
        v = prefilter(v)
    m32 = np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(4, 4))
    out = output if output is not None else np.zeros(v.shape, dtype=np.float32)
    rc = lib().vt_oracle_affine(v, *v.shape, m32.ravel(), out, code, flags)
    if rc:
        raise RuntimeError(f'vt_oracle_affine failed ({rc})')
    return out


def affine_ex(src: np.ndarray, m64: np.ndarray, interpolation, out_shape, plane0=0, global_depth=None,
              out_plane0=0, flags: int = 0) -> np.ndarray:
    """Generalised form (slab windows, output shape != input shape, float64 matrix)."""
    code = _interp_code(interpolation)
    v = np.ascontiguousarray(src, dtype=np.float32)
    gD = v.shape[0] if global_depth is None else global_depth
    out = np.zeros(tuple(out_shape), dtype=np.float32)
    m = np.ascontiguousarray(np.asarray(m64, dtype=np.float64).reshape(16))
    rc = lib().vt_oracle_affine_ex(v, *v.shape, plane0, gD, out, *out.shape, out_plane0, m, code, flags)
    if rc:
        raise RuntimeError(f'vt_oracle_affine_ex failed ({rc})')
    return out


def num_threads() -> int:
    return lib().vt_oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().vt_oracle_set_num_threads(n)
