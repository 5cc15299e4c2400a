import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
from voltools_amd import _native
lib = _native.load()
n = 512
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
out = np.zeros((n, n, n), np.float32)
c = np.divide((n, n, n), 2)
lib.vt_affine_oneshot.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
for name, m in (('general', vt.utils.transform_matrix(rotation=(10, 20, 30), rotation_order='sxyz', center=c)),
                ('axis0', vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=c))):
    m = np.ascontiguousarray(m, dtype=np.float32)
    for flags in (0, 128, 4):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            rc = lib.vt_affine_oneshot(0, data.ctypes.data, n, n, n, 0, m.ctypes.data, out.ctypes.data, flags, None)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0
        print(name, 'flags', flags, ' '.join(f'{t:.2f}' for t in ts), 'ms')
