"""Large non-cubic shapes: every default path against the direct kernel (same library, independent indexing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voltools_amd as vt
from voltools_amd import _native
ok = True
for shape in ((48, 1500, 2100), (1500, 48, 2100), (2100, 1500, 48), (40, 40, 5000), (3000, 3000, 12), (12, 3000, 3000)):
    vol = np.random.RandomState(3).random_sample(shape).astype(np.float32)
    c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    for interp in ('linear', 'filt_bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        out = vt.empty(shape, device='gpu:0')
        for name, m in (('axis0', vt.utils.transform_matrix(rotation=(0, 37, 0), translation=(1.5, -2, 3), center=c)),
                        ('axis0_100', vt.utils.transform_matrix(rotation=(0, 100, 0), center=c)),
                        ('axis1', vt.utils.transform_matrix(rotation=(0, 37, 0), rotation_order='sxyz', center=c)),
                        ('axis2', vt.utils.transform_matrix(rotation=(0, 0, 37), rotation_order='sxyz', center=c)),
                        ('general', vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=c))):
            sv.affine(m, output=out); a = out.get(); k = sv.info().last_kernel
            sv.affine(m, output=out, _flags=_native.FORCE_DIRECT); b = out.get()
            err = float(np.abs(a - b).max())
            good = err <= (2e-6 if interp == 'linear' else 1e-5)
            ok = ok and good
            print(shape, interp, name, 'kernel', k, f'err {err:.2e}', 'ok' if good else 'FAIL', flush=True)
        pr = sv.projection(vt.utils.transform_matrix(rotation=(0, 37, 0), center=c))
        sv.close(); out.free()
print('ALL OK' if ok else 'FAILURES')
