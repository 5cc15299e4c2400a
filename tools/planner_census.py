"""Which kernel family serves what: histogram of vt_volume_info.last_kernel over shapes >= 64^3 and matrix classes (default planner,
no flags).  Decides whether the round-1 families (3 tiled axis-0-separable, 4 plain marching, 5 plane-pair marching) are ever chosen.
    python3 tools/planner_census.py"""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt

rs = np.random.RandomState(0)
shapes = [(64, 64, 64), (96, 80, 72), (128, 128, 128), (100, 300, 50), (256, 256, 256), (200, 320, 180), (40, 512, 512), (512, 64, 640), (384, 384, 384)]
hist = collections.defaultdict(collections.Counter)
examples = {}
for shape in shapes:
    vol = rs.random_sample(shape).astype(np.float32)
    c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    out = vt.empty(shape, device='gpu:0')
    for interp in ('linear', 'bspline'):
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        cases = []
        for a in (0, 7, 30, 45, 60, 90, 133, 180):
            cases.append(('rot axis0', vt.utils.transform_matrix(rotation=(0, a, 0), center=c)))
            cases.append(('rot axis1', vt.utils.transform_matrix(rotation=(a, 0, 0), rotation_order='sxyz', center=c) if False else vt.utils.transform_matrix(rotation=(0, a, 0), rotation_order='ryxy', center=c)))
            cases.append(('rot axis2', vt.utils.transform_matrix(rotation=(a, 0, 0), rotation_order='rzxz', center=c)))
        for s in (0.2, 0.35, 0.5, 0.8, 1.25, 2.0, 4.0):
            cases.append((f'inplane scale', vt.utils.transform_matrix(scale=(1.0, s, s), rotation=(0, 20, 0), center=c)))
            cases.append((f'uniform scale', vt.utils.transform_matrix(scale=(s, s, s), center=c)))
        for t in ((3.5, 0, 0), (0, 10.25, -3), (0.5, 0.5, 0.5)):
            cases.append(('translation', vt.utils.transform_matrix(translation=t, center=c)))
        cases.append(('shear', vt.utils.transform_matrix(shear=(0.2, 0.1, 0.0), center=c)))
        for r in rs.uniform(-180, 180, (6, 3)):
            cases.append(('general rot', vt.utils.transform_matrix(rotation=tuple(r), rotation_order='sxyz', center=c)))
        for name, m in cases:
            sv.affine(m, output=out)
            k = int(sv.info().last_kernel)
            hist[(interp, name)][k] += 1
            examples.setdefault((interp, k), (shape, name))
        sv.close()
    out.free()
for key in sorted(hist):
    print(f'{key[0]:8s} {key[1]:14s}: ' + '  '.join(f'kernel {k}: {n}' for k, n in sorted(hist[key].items())))
print('first example per (interp, kernel):')
for key in sorted(examples):
    print(f'  {key}: {examples[key]}')
