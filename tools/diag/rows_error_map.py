"""Where does the row kernel (kind 7) differ from affine_direct?  Error statistics by pixel-in-tile, w run and d (debugging aid)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np
import voltools_amd as vt
from voltools_amd import _native
import itertools
for shape, ang in (((128, 128, 128), 5.0), ((128, 128, 128), 12.0), ((128, 128, 128), 20.0), ((128, 128, 128), 33.0), ((128, 128, 128), 80.0)):
  if True:
    vol = np.random.RandomState(51).random_sample(shape).astype(np.float32)
    c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    m = vt.utils.transform_matrix(rotation=(0, 0, ang), rotation_order='sxyz', center=c)
    for interp in ('linear', 'bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        got0 = sv.affine(m, _flags=_native.FORCE_TILED); sv.synchronize()
        got1 = sv.affine(m, _flags=_native.FORCE_TILED)
        got = sv.affine(m, _flags=_native.FORCE_TILED); k = sv.info().last_kernel
        print('   first call == third call:', np.array_equal(got0, got), ' second == third:', np.array_equal(got1, got))
        ref = sv.affine(m, _flags=_native.FORCE_DIRECT)
        err = np.abs(got - ref)
        print(shape, ang, list(sv.info().last_lds_dims), interp, 'kernel', k, 'max err', err.max(), 'bad voxels', int((err > 1e-6).sum()), 'of', err.size)
        if err.max() > 1e-6:
            bad = err > 1e-6
            print('  by h % 8:', [int(bad[:, i::8, :].sum()) for i in range(8)])
            print('  by d % 4:', [int(bad[i::4].sum()) for i in range(4)])
            print('  by w // 64:', [int(bad[:, :, 64 * i:64 * i + 64].sum()) for i in range(2)], ' w % 4:', [int(bad[:, :, i::4].sum()) for i in range(4)])
            d, h, w = np.argwhere(bad)[0]
            print('  first bad voxel', (d, h, w), 'got', got[d, h, w], 'ref', ref[d, h, w], ' ratio', got[d, h, w] / max(ref[d, h, w], 1e-30))
            print('  got == 0 where bad:', int((got[bad] == 0).sum()), ' ref == 0 where bad:', int((ref[bad] == 0).sum()))
        sv.close()
