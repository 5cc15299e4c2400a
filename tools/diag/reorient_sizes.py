"""General rotations on the axis-permuted copy (default from the fourth request on) against the plain copy (NO_REORIENT), by size.
usage: python3 tools/diag/reorient_sizes.py [size ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt  # noqa: E402
from voltools_amd import _native as N  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [160, 192, 224, 256, 320]
rs = np.random.RandomState(1)
rots = rs.uniform(-180, 180, (16, 3))
for n in sizes:
    shape = (n, n, n)
    c = np.divide(shape, 2)
    mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=c) for r in rots]
    vol = np.random.RandomState(0).random_sample(shape).astype(np.float32)
    out = vt.empty(shape, device='gpu:0')
    for interp in ('linear', 'filt_bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        res = {}
        for name, fl in (('reoriented', 0), ('plain', N.NO_REORIENT), ('reoriented again', 0)):
            for _ in range(5):
                for m in mats:
                    sv.affine(m, output=out, _flags=fl)
            sv.synchronize(); sv.timer_start()
            for _ in range(3):
                for m in mats:
                    sv.affine(m, output=out, _flags=fl)
            res[name] = sv.timer_stop() / (3 * len(mats))
        i = sv.info()
        print('%d^3 %-13s reoriented %.4f / %.4f ms   plain copy only %.4f   (copies built %d, last kernel %d)'
              % (n, interp, res['reoriented'], res['reoriented again'], res['plain'], i.copies_built, i.last_kernel), flush=True)
        sv.close()
    out.free()
