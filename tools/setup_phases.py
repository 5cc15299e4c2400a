"""Cycle counts of the plane-quad kernel's set-up phases (experiment build, VT_EXP_NOLOOP=1: thread 0 of every workgroup leaves four
clock64() differences in the output buffer): geometry | span atomics | placement by wave 0 | vector offsets.
    VT_LIB=.../lib_b/libvoltools_hip.so VT_EXP_NOLOOP=1 python3 tools/setup_phases.py --size 512 --interp filt_bspline --angle 30"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--interp', default='filt_bspline')
ap.add_argument('--angle', type=float, default=30.0)
a = ap.parse_args()
n = a.size
vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
sv = vt.StaticVolume(vol, interpolation=a.interp, device='gpu:0')
out = vt.empty((n, n, n), device='gpu:0')
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
m = vt.utils.transform_matrix(rotation=(0, a.angle, 0), rotation_order='rzxz', center=c)
for _ in range(3):
    sv.affine(m, output=out)
sv.synchronize()
info = sv.info()
g = int(info.last_grid)
host = out.get() if hasattr(out, 'get') else np.asarray(out.cpu())
st = host.reshape(-1)[:4 * g].reshape(g, 4)
print(f'{a.interp} {n}^3 angle {a.angle}: kernel {info.last_kernel} grid {g}; mean cycles per workgroup: geometry {st[:,0].mean():.0f}, '
      f'atomics {st[:,1].mean():.0f}, placement {st[:,2].mean():.0f}, offsets {st[:,3].mean():.0f}; total {st.sum(1).mean():.0f} '
      f'(median {np.median(st.sum(1)):.0f})')
