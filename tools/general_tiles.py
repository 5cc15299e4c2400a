"""Per-matrix timing of the general-matrix kernel families on the reference's 100 random rotations: the planner's choice against
each family forced (packed footprints, lane blocks, bounding boxes) -- what a better dispatch rule could gain.
    python3 tools/general_tiles.py [size] [interp]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt
from voltools_amd import _native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
interp = sys.argv[2] if len(sys.argv) > 2 else 'linear'
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rs.uniform(-180, 180, (100, 3))]
out = vt.zeros((n, n, n), device='gpu:0')
variants = {'default': 0, 'packed': _native.FORCE_PACKED, 'no_block': _native.NO_BLOCK, 'boxes': _native.NO_BLOCK | _native.NO_PACKED}
sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
res = {}
for name, flags in variants.items():
    ts, ks = [], []
    for m in mats:
        sv.affine(m, output=out, _flags=flags)
        sv.synchronize()
        sv.timer_start()
        for _ in range(3):
            sv.affine(m, output=out, _flags=flags)
        ts.append(sv.timer_stop() / 3)
        ks.append(sv.info().last_kernel)
    res[name] = (np.array(ts), np.array(ks))
sv.close()
d, dk = res['default']
print(f'{n}^3 {interp}: ' + '  '.join(f'{k} {v[0].mean():.4f} ms (kernels {dict(zip(*np.unique(v[1], return_counts=True)))})' for k, v in res.items()))
allt = np.stack([v[0] for v in res.values()])
print(f'per-matrix best of all variants: {allt.min(axis=0).mean():.4f} ms')
for k in np.unique(dk):
    sel = dk == k
    print(f'where the default took kernel {k} ({sel.sum()} matrices): ' + '  '.join(f'{name} {v[0][sel].mean():.4f}' for name, v in res.items()))
