"""100 random `sxyz` rotations (the reference's benchmark protocol, tests/benchmark.py:52-54) on a resident volume, device output:
mean ms per transform with the default planner and with the lane-block kernel disabled (VT_NO_BLOCK flag), in one process.
    python3 tools/general_ab.py [size ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt
from voltools_amd import _native

for n in [int(a) for a in sys.argv[1:]] or [250, 384, 512]:
    rs = np.random.RandomState(1)
    data = rs.random_sample((n, n, n)).astype(np.float32)
    mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rs.uniform(-180, 180, (100, 3))]
    for interp in ('linear', 'filt_bspline'):
        sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
        out = vt.zeros((n, n, n), device='gpu:0')
        row = []
        for flags in (0, _native.NO_BLOCK, 0, _native.NO_BLOCK):
            for m in mats[:5]:
                sv.affine(m, output=out, _flags=flags)
            sv.synchronize()
            sv.timer_start()
            for m in mats:
                sv.affine(m, output=out, _flags=flags)
            row.append(sv.timer_stop() / len(mats))
        k = sv.info().last_kernel
        print(f'{n}^3 {interp}: default {min(row[0], row[2]):.4f} ms, without lane-block kernel {min(row[1], row[3]):.4f} ms  '
              f'({8.0 * n ** 3 / min(row[0], row[2]) / 1e6:.0f} GB/s algorithmic)', flush=True)
        sv.close()
