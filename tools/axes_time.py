import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import voltools_amd as vt
n = 512
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
out = vt.zeros((n, n, n), device='gpu:0')
for interp in ('linear', 'filt_bspline'):
    sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
    for ax in range(3):
        mats = []
        for a in range(5, 180, 7):
            r = [0, 0, 0]; r[ax] = float(a)
            mats.append(vt.utils.transform_matrix(rotation=tuple(r), rotation_order='sxyz', center=np.divide((n, n, n), 2)))
        for m in mats[:3]: sv.affine(m, output=out)
        sv.synchronize(); sv.timer_start()
        for _ in range(2):
            for m in mats: sv.affine(m, output=out)
        t = sv.timer_stop() / (2 * len(mats))
        i = sv.info()
        print(interp, 'sxyz axis', ax, f'{t:.4f} ms', 'kernel', i.last_kernel, flush=True)
    sv.close()
