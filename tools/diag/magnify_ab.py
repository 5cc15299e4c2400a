"""Magnifying maps: the planner's default against the direct kernel (FORCE_DIRECT).  usage: python3 tools/diag/magnify_ab.py [size]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt  # noqa: E402
from voltools_amd import _native as N  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
T = vt.utils.transform_matrix
cases = {}
for s in (1.5, 2.0, 2.25, 2.5, 2.75, 3.0, 3.5):
    cases['scale %.2f' % s] = T(scale=(s, s, s), center=c)
    cases['rot(25,-40,70) scale %.2f' % s] = T(rotation=(25, -40, 70), rotation_order='sxyz', scale=(s, s, s), center=c)
    cases['scale (%.2f, %.2f, 1)' % (s, s)] = T(scale=(s, s, 1.0), center=c)
    cases['rot axis0 33 scale %.2f' % s] = T(rotation=(0, 33, 0), scale=(s, s, s), center=c)
cases['scale (3,1,1)'] = T(scale=(3.0, 1.0, 1.0), center=c)
cases['scale (1,1,3)'] = T(scale=(1.0, 1.0, 3.0), center=c)
vol = np.random.RandomState(0).random_sample(shape).astype(np.float32)
out = vt.empty(shape, device='gpu:0')
for interp in ('linear', 'filt_bspline'):
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for name, m in cases.items():
        row = []
        for fl in (0, N.FORCE_DIRECT, N.NO_QUAD):
            for _ in range(5):
                sv.affine(m, output=out, _flags=fl)
            sv.synchronize(); sv.timer_start()
            for _ in range(10):
                sv.affine(m, output=out, _flags=fl)
            i = sv.info()
            row.append((sv.timer_stop() / 10, int(i.last_kernel), tuple(i.last_tile), int(i.last_lds_bytes)))
        t = row[0][2]
        bpv = row[0][3] / max(1, t[0] * t[1] * t[2])
        print('%-13s %-28s default %.4f (k%d)  direct %.4f  no_quad %.4f (k%d)%s  tile %s lds %d bpv %.0f'
              % (interp, name, row[0][0], row[0][1], row[1][0], row[2][0], row[2][1], '   <<<' if row[1][0] < 0.95 * row[0][0] else '',
                 t, row[0][3], bpv), flush=True)
    sv.close()
