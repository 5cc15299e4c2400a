"""Cubic general matrices: the lane-block kernel (kind 9, the default from 240^3 on) against the bounding-box kernel (kind 2, NO_BLOCK) by how
compact the box kernel's boxes are (staged bytes per voxel).  usage: python3 tools/diag/block_vs_box.py [size ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt  # noqa: E402
from voltools_amd import _native as N  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 384, 512]
T = vt.utils.transform_matrix
for n in sizes:
    shape = (n, n, n)
    c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    cases = {}
    for a in (2, 5, 10, 20, 33, 45):
        cases['rot(%d,%d,%d)' % (a, a, a)] = T(rotation=(a, a, a), rotation_order='sxyz', center=c)
    cases['rot(25,-40,70)'] = T(rotation=(25, -40, 70), rotation_order='sxyz', center=c)
    cases['shear'] = T(shear=(0.1, -0.05, 0.2), center=c)
    cases['shear_big'] = T(shear=(0.4, -0.3, 0.5), center=c)
    cases['magnify(0.4,0.5,0.3)'] = T(scale=(0.4, 0.5, 0.3), center=c)
    cases['scale 0.8'] = T(scale=(0.8, 0.8, 0.8), center=c)
    cases['scale 1.2 rot(10,20,30)'] = T(rotation=(10, 20, 30), scale=(1.2, 1.2, 1.2), rotation_order='sxyz', center=c)
    vol = np.random.RandomState(0).random_sample(shape).astype(np.float32)
    out = vt.empty(shape, device='gpu:0')
    for interp in ('filt_bspline', 'linear'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for name, m in cases.items():
            row = []
            for fl in (0, N.NO_BLOCK | N.NO_PACKED, 0):
                for _ in range(5):
                    sv.affine(m, output=out, _flags=fl)
                sv.synchronize(); sv.timer_start()
                for _ in range(10):
                    sv.affine(m, output=out, _flags=fl)
                ms = sv.timer_stop() / 10
                i = sv.info()
                t = tuple(i.last_tile)
                row.append((ms, int(i.last_kernel), t, int(i.last_lds_bytes)))
            d = min(row[0], row[2])
            b = row[1]
            bpv = b[3] / max(1, b[2][0] * b[2][1] * b[2][2])
            print('%d^3 %-13s %-26s default %.4f (k%d)   boxes %.4f (k%d, tile %s, %d B, %.0f B/voxel)%s'
                  % (n, interp, name, d[0], d[1], b[0], b[1], b[2], b[3], bpv, '   <<< boxes' if b[0] < 0.95 * d[0] else ''), flush=True)
        sv.close()
    out.free()
