import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
shape = (256, 192, 320)
vol = np.random.RandomState(21).random_sample(shape).astype(np.float32)
c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
sv = vt.StaticVolume(vol, interpolation='linear', device='gpu:0')
for name, m in (('rot', vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(2.5, 1.0, -3.0), center=c)),
                ('shift', vt.utils.translation_matrix((-7.25, 0.5, 0.0))), ('eye', np.eye(4, dtype=np.float32)),
                ('big', vt.utils.translation_matrix((300.0, 0, 0)))):
    want = sv.affine(m)
    got = vt.affine(vol, m, interpolation='linear', device='gpu')
    d = np.abs(got - want)
    print(name, 'max diff', d.max(), 'n diff', int((d > 0).sum()), 'planes', np.unique(np.nonzero(d.max(axis=(1, 2)))[0])[:20])
n = 512
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
m = vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=np.divide((n, n, n), 2))
for _ in range(2):
    vt.affine(data, m, interpolation='linear', device='gpu')
t0 = time.perf_counter()
for _ in range(5):
    vt.affine(data, m, interpolation='linear', device='gpu')
print('512 axis0 linear', os.environ.get('VT_PIPE_NCH'), os.environ.get('VT_ONESHOT_SEQ'), (time.perf_counter() - t0) / 5 * 1e3, 'ms')
