"""Routing audit: for every matrix class of the parity tests (plus a few of this round's), the planner's default against every
alternative a diagnostic flag can force, timed on one handle -- a default that is not the fastest by more than 5 % is printed with '<<<'.
Found this way in round 5: cubic launches with a fractional axis-2 offset on the row kernel (0.60 ms against 0.44 on the exchange path).
usage: python3 tools/routing_audit.py [size] [interp ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import voltools_amd as vt  # noqa: E402
from voltools_amd import _native as N  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
interps = sys.argv[2:] or ['linear', 'filt_bspline']
shape = (n, n, n)
c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
T = vt.utils.transform_matrix
cases = {
    'rot_axis0_33': T(rotation=(0, 33, 0), center=c),
    'rot_axis0_100_shift': T(rotation=(0, 100, 0), translation=(0.5, -1.25, 2.0), center=c),
    'rot_axis0_33_zfrac': None,
    'rot_axis1_33': T(rotation=(0, 33, 0), rotation_order='sxyz', center=c),
    'rot_axis1_120_shift': T(rotation=(0, -120, 0), rotation_order='sxyz', translation=(0.5, 2.0, -1.25), center=c),
    'rot_axis2_33': T(rotation=(0, 0, 33), rotation_order='sxyz', center=c),
    'rot_axis2_33_w+2': None, 'rot_axis2_33_w+0.5': None,
    'rot_general': T(rotation=(25, -40, 70), rotation_order='sxyz', center=c),
    'rot_general_b': T(rotation=(-100, 15, 160), rotation_order='sxyz', center=c),
    'rot_scale_shift': T(rotation=(10, 20, 30), scale=(1.1, 0.9, 1.25), translation=(2, -3, 1.5), rotation_order='sxyz', center=c),
    'shear': T(shear=(0.1, -0.05, 0.2), center=c),
    'magnify3': T(scale=(3.0, 3.0, 3.0), center=c),
    'minify': T(scale=(0.4, 0.5, 0.3), center=c),
    'shift_int': vt.utils.translation_matrix((3, -2, 5)),
    'shift_frac': vt.utils.translation_matrix((0.5, -1.25, 2.75)),
    'mirror': T(scale=(-1.0, 1.0, -1.0), center=c),
}
m = cases['rot_axis0_33'].copy(); m[0, 3] += 0.4; cases['rot_axis0_33_zfrac'] = m
m = cases['rot_axis2_33'].copy(); m[2, 3] += 2.0; cases['rot_axis2_33_w+2'] = m
m = cases['rot_axis2_33'].copy(); m[2, 3] += 0.5; cases['rot_axis2_33_w+0.5'] = m
variants = [('default', 0), ('NO_ROWS', N.NO_ROWS), ('NO_QUAD', N.NO_QUAD), ('NO_ZFIR', N.NO_ZFIR), ('NO_RSWAP', N.NO_RSWAP),
            ('FORCE_XSWAP', N.FORCE_XSWAP), ('NO_ZSEP', N.NO_ZSEP), ('NO_BLOCK', N.NO_BLOCK), ('NO_PACKED', N.NO_PACKED),
            ('NO_BLOCK|NO_PACKED', N.NO_BLOCK | N.NO_PACKED), ('NO_REORIENT', N.NO_REORIENT), ('FORCE_DIRECT', N.FORCE_DIRECT),
            ('default again', 0)]      # (the first measurement of a case also warms the clocks and builds lazily created copies)
rs = np.random.RandomState(0)
vol = rs.random_sample(shape).astype(np.float32)
out = vt.empty(shape, device='gpu:0')
for interp in interps:
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for name, m in cases.items():
        res = []
        for vname, fl in variants:
            for _ in range(5):                       # (reoriented copies are built at the fourth request)
                sv.affine(m, output=out, _flags=fl)
            sv.synchronize()
            sv.timer_start()
            for _ in range(10):
                sv.affine(m, output=out, _flags=fl)
            ms = sv.timer_stop() / 10
            res.append((vname, ms, int(sv.info().last_kernel)))
        d = min(res[0], res[-1], key=lambda r: r[1])
        seen = {}
        for vname, ms, k in res:
            seen.setdefault((k, round(ms, 2)), (vname, ms, k))
        best = min(res[:-1], key=lambda r: r[1])
        flag = '  <<< ' + best[0] if best[1] < 0.95 * d[1] else ''
        alts = ' '.join('%s=%.3f(k%d)' % r for r in res[1:-1] if abs(r[1] - d[1]) > 0.02 * d[1] or r[2] != d[2])
        print('%-13s %-22s default %.4f ms (k%d)%s | %s' % (interp, name, d[1], d[2], flag, alts), flush=True)
    sv.close()
