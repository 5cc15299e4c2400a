import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
n = 100
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
rot = rs.uniform(-180, 180, (100, 3))
c = np.divide((n, n, n), 2)
mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=c) for r in rot]
def run(tag, ms):
    for m in ms[:3]:
        vt.affine(data, m, interpolation='linear', device='gpu')
    t0 = time.perf_counter()
    ts = []
    for m in ms:
        t1 = time.perf_counter()
        vt.affine(data, m, interpolation='linear', device='gpu')
        ts.append((time.perf_counter() - t1) * 1e3)
    ts = np.array(ts)
    print(tag, f'mean {ts.mean():.3f} median {np.median(ts):.3f} min {ts.min():.3f} max {ts.max():.3f} ms')
run('fixed matrix     ', [mats[0]] * 100)
run('random matrices  ', mats)
sv = vt.StaticVolume(data, interpolation='linear', device='gpu:0')
out = vt.zeros((n, n, n), device='gpu:0')
for m in mats:
    sv.affine(m, output=out)
sv.synchronize()
run('random, after sv  ', mats)
for m in mats:
    sv.affine(m)
run('random, after sv->numpy', mats)
