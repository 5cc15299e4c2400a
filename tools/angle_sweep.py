"""Per-angle kernel time of the resident transform (README sweep family): python3 tools/angle_sweep.py --interp filt_bspline"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt
ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--interp', default='filt_bspline')
ap.add_argument('--step', type=int, default=3)
args = ap.parse_args()
n = args.size
if n >= 768:                                   # generate on the device: 4 GiB of host random numbers is most of a short run
    import torch
    vol = torch.rand((n, n, n), dtype=torch.float32, device='cuda:0')
else:
    vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
sv = vt.StaticVolume(vol, interpolation=args.interp, device='gpu:0')
out = vt.empty((n, n, n), device='gpu:0')
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
res = []
for ang in range(0, 180, args.step):
    m = vt.utils.transform_matrix(rotation=(0, ang, 0), center=c)
    sv.affine(m, output=out)
    sv.synchronize()
    sv.timer_start()
    for _ in range(5):
        sv.affine(m, output=out)
    ms = sv.timer_stop() / 5
    info = sv.info()
    res.append((ang, ms, tuple(info.last_lds_dims), info.last_lds_bytes))
print(args.interp, n, 'mean ms', np.mean([r[1] for r in res]), 'max', max(r[1] for r in res))
print(' '.join(f'{a}:{ms:.3f}/{d[2]}' for a, ms, d, b in res))
