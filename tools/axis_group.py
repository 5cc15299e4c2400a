"""Per-matrix time of the trilinear general-matrix kernel on the reference's 100 random rotations, grouped by the output axis along
which the source x coordinate moves most (the axis whose neighbouring tiles share source cache lines).   python3 tools/axis_group.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import voltools_amd as vt
n = 512
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rs.uniform(-180, 180, (100, 3))]
out = vt.zeros((n, n, n), device='gpu:0')
sv = vt.StaticVolume(data, interpolation='linear', device='gpu:0')
ts = []
for m in mats:
    sv.affine(m, output=out); sv.synchronize(); sv.timer_start()
    for _ in range(3): sv.affine(m, output=out)
    ts.append(sv.timer_stop() / 3)
ts = np.array(ts)
k = np.array([int(np.argmax(np.abs(np.asarray(m)[2, :3]))) for m in mats])
mx = np.array([float(np.max(np.abs(np.asarray(m)[2, :3]))) for m in mats])
for a in range(3):
    sel = k == a
    print('source x follows output axis', a, ':', sel.sum(), 'matrices, mean', round(float(ts[sel].mean()), 4), 'ms; |m| > 0.8:', round(float(ts[sel & (mx > 0.8)].mean()), 4) if (sel & (mx > 0.8)).any() else None)
print('all', round(float(ts.mean()), 4))
