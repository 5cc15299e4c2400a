import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
n = 1024
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
m = vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=np.divide((n, n, n), 2))
for interp in ('linear', 'filt_bspline'):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = vt.affine(data, m, interpolation=interp, device='gpu')
        ts.append((time.perf_counter() - t0) * 1e3)
    print('1024 axis0', interp, os.environ.get('VT_ONESHOT_SEQ'), ' '.join(f'{t:.1f}' for t in ts), 'ms', flush=True)
    sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
    want = sv.affine(m)
    print('   max diff vs resident', float(np.abs(r - want).max()), flush=True)
    sv.close(); del want
vt.free_cached_memory()
