"""Per-matrix launch time over the 100 random `sxyz` rotations of the reference's protocol (tests/benchmark.py:52-54): which rotations are the
slow ones?   python3 tools/general_per_matrix.py [size] [interp] > gpurun_out/per_matrix.txt      (columns: index, ms, kernel, LDS bytes, three angles)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
interp = sys.argv[2] if len(sys.argv) > 2 else 'filt_bspline'
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
rots = rs.uniform(-180, 180, (100, 3))
mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rots]
sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
out = vt.zeros((n, n, n), device='gpu:0')
for m in mats[:5]:
    sv.affine(m, output=out)
sv.synchronize()
for i, m in enumerate(mats):
    sv.affine(m, output=out)
    sv.timer_start()
    for _ in range(10):
        sv.affine(m, output=out)
    ms = sv.timer_stop() / 10
    info = sv.info()
    print(i, f'{ms:.4f}', info.last_kernel, info.last_lds_bytes, *[f'{a:.3f}' for a in rots[i]], flush=True)
