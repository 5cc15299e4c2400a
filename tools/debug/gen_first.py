import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
n = 512
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
c = np.divide((n, n, n), 2)
mg = vt.utils.transform_matrix(rotation=(10, 20, 30), rotation_order='sxyz', center=c)
ma = vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=c)
out = vt.empty((n, n, n), device='gpu:0')
for interp in ('linear', 'filt_bspline'):
    for name, m in (('general', mg), ('axis0', ma)):
        t0 = time.perf_counter(); sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0'); t1 = time.perf_counter()
        ts = []
        for _ in range(4):
            t = time.perf_counter(); sv.affine(m, output=out); sv.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        t2 = time.perf_counter(); r = sv.affine(m); t3 = time.perf_counter()
        print(interp, name, f'create {(t1-t0)*1e3:.2f} ms; affine(output=) calls ' + ' '.join(f'{x:.2f}' for x in ts) + f' ms; ->numpy {(t3-t2)*1e3:.2f} ms; kernel', sv.info().last_kernel)
        sv.close()
