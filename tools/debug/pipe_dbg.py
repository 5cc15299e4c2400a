import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
shape = (256, 192, 320)
vol = np.random.RandomState(21).random_sample(shape).astype(np.float32)
c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
for interp in ('linear', 'bspline', 'filt_bspline'):
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for name, m in (('rot', vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(2.5, 1.0, -3.0), center=c)),
                    ('shift', vt.utils.translation_matrix((-7.25, 0.5, 0.0))), ('eye', np.eye(4, dtype=np.float32)),
                    ('big', vt.utils.translation_matrix((300.0, 0, 0))), ('neg', vt.utils.translation_matrix((-40.0, 3, 0)))):
        want = sv.affine(m)
        got = vt.affine(vol, m, interpolation=interp, device='gpu')
        d = np.abs(got - want)
        print(interp, name, 'max diff', d.max(), 'n diff', int((d > 0).sum()))
    sv.close()
n = 512
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
m = vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=np.divide((n, n, n), 2))
for interp in ('linear', 'filt_bspline'):
    for _ in range(3):
        vt.affine(data, m, interpolation=interp, device='gpu')
    t0 = time.perf_counter()
    for _ in range(5):
        vt.affine(data, m, interpolation=interp, device='gpu')
    print('512 axis0', interp, os.environ.get('VT_ONESHOT_SEQ'), (time.perf_counter() - t0) / 5 * 1e3, 'ms')
