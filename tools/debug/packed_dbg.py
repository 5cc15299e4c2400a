"""Debug helper: packed-footprint kernel (kind 6) vs the direct gather, with mismatch locations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
import voltools_amd as vt
from voltools_amd import _native
from test_gpu_parity import MATRICES, rand_vol

for shape in [(70, 66, 72), (33, 47, 50)]:
    vol = rand_vol(shape, 1)
    for interp in ('linear', 'bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for name in ('identity', 'shift_frac', 'rot_general', 'shear', 'magnify3', 'minify', 'mirror', 'rot_scale_shift'):
            m = MATRICES[name](shape)
            want = sv.affine(m, _flags=_native.FORCE_DIRECT)
            got = np.full(shape, -777.0, np.float32)
            sv.affine(m, output=got, keep_outside=False, _flags=_native.FORCE_TILED | _native.NO_ZSEP)
            info = sv.info()
            bad = ~(np.abs(got - want) <= 2e-6)
            idx = np.argwhere(bad)
            print(shape, interp, name, 'kernel', info.last_kernel, 'tile', list(info.last_tile), 'lds', list(info.last_lds_dims), info.last_lds_bytes,
                  'grid', info.last_grid, 'bad', int(bad.sum()), 'nan', int(np.isnan(got).sum()), 'unwritten', int((got == -777).sum()))
            if len(idx):
                print('   first', idx[:6].tolist(), 'min', idx.min(0).tolist(), 'max', idx.max(0).tolist())
                for i in idx[:4]:
                    print('   ', tuple(i), got[tuple(i)], want[tuple(i)])
        sv.close()
