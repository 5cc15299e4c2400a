"""Projection timing: fused (axis-0 rotation) vs transform-then-sum.  python3 tools/proj_time.py --size 512 --interp filt_bspline"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voltools_amd as vt
from voltools_amd import _native
ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--interp', default='filt_bspline')
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
n = args.size
vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
sv = vt.StaticVolume(vol, interpolation=args.interp, device='gpu:0')
out = vt.empty((n, n), device='gpu:0')
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
for label, rot, order, flags in (('fused, rotation about axis 0', (30, 0, 0), 'sxyz', 0),
                                 ('unfused, same matrix', (30, 0, 0), 'sxyz', _native.NO_ZSEP),
                                 ('general rotation', (25, -40, 70), 'sxyz', 0)):
    m = vt.utils.transform_matrix(rotation=rot, rotation_order=order, center=c)
    for _ in range(3):
        sv.projection(m, output=out, _flags=flags)
    sv.synchronize()
    sv.timer_start()
    for _ in range(args.iters):
        sv.projection(m, output=out, _flags=flags)
    ms = sv.timer_stop() / args.iters
    print(f'{args.interp} {n}^3 projection [{label}]: {ms:.4f} ms, {n ** 3 / ms / 1e6:.1f} Gvox/s, '
          f'{4.0 * n ** 3 / ms / 1e6:.0f} GB/s of source read, kernel={sv.info().last_kernel}')
