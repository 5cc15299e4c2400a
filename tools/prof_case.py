"""Run one resident-transform configuration repeatedly (for rocprofv3 kernel traces / PMC passes).

    python3 tools/prof_case.py --size 512 --interp linear --angle 45 --iters 10 [--order rzxz] [--general]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--interp', default='linear')
ap.add_argument('--angle', type=float, default=45.0)
ap.add_argument('--axis1', action='store_true', help='rotation by --angle about array axis 1 (sxyz (0, angle, 0))')
ap.add_argument('--axis2', action='store_true', help='rotation by --angle about array axis 2 (sxyz (0, 0, angle))')
ap.add_argument('--sweep', type=float, default=0.0, help='README sweep: rotate((0, i, 0)) for i = 0, step, 2 step, ... < 180 instead of one angle')
ap.add_argument('--general', action='store_true', help='general 3-D rotation (25,-40,70) sxyz instead of in-plane')
ap.add_argument('--case', default='', help='named matrix from tests/test_gpu_parity.py MATRICES (overrides --angle/--general)')
ap.add_argument('--random100', action='store_true', help="the 100 random sxyz rotations of the reference's protocol (tests/benchmark.py:52-54), one after the other")
ap.add_argument('--shift2', type=float, default=0.0, help='added to the axis-2 offset of the matrix (row kernel: integer / fractional offsets)')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--flags', type=int, default=0)
args = ap.parse_args()

n = args.size
if n >= 768:                                   # 4 GiB of host random numbers per process is most of a short run: generate on the device
    import torch
    g = torch.Generator(device='cuda:0'); g.manual_seed(0)
    vol = torch.rand((n, n, n), dtype=torch.float32, device='cuda:0', generator=g)
else:
    vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
sv = vt.StaticVolume(vol, interpolation=args.interp, device='gpu:0')
out = vt.empty((n, n, n), device='gpu:0')
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
if args.general:
    m = vt.utils.transform_matrix(rotation=(25, -40, 70), rotation_order='sxyz', center=c)
else:
    m = vt.utils.transform_matrix(rotation=(0, args.angle, 0), rotation_order='rzxz', center=c)
if args.axis1:
    m = vt.utils.transform_matrix(rotation=(0, args.angle, 0), rotation_order='sxyz', center=c)
if args.axis2:
    m = vt.utils.transform_matrix(rotation=(0, 0, args.angle), rotation_order='sxyz', center=c)
    m[2, 3] += args.shift2
if args.case:
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from test_gpu_parity import MATRICES
    m = MATRICES[args.case]((n, n, n))
mats = [m]
if args.random100:
    rs_ = np.random.RandomState(1)
    rs_.random_sample(n * n * n) if n < 768 else None
    mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rs_.uniform(-180, 180, (100, 3))]
    args.sweep = 1.0
if args.sweep > 0 and not args.random100:
    mats = [vt.utils.transform_matrix(rotation=(0, float(a), 0), rotation_order='rzxz', center=c) for a in np.arange(0.0, 180.0, args.sweep)]
for mm in (mats if args.sweep > 0 else [m] * 3):
    sv.affine(mm, output=out, _flags=args.flags)        # (a sweep runs once untimed: the lazily built resident copies exist afterwards)
sv.synchronize()
sv.timer_start()
for i in range(args.iters):
    sv.affine(mats[i % len(mats)], output=out, _flags=args.flags)
ms = sv.timer_stop() / args.iters
info = sv.info()
print(f'{args.interp} {n}^3 angle={args.angle} general={args.general} case={args.case} kernel={info.last_kernel}: {ms:.4f} ms/launch, '
      f'{n ** 3 / ms / 1e6:.1f} Gvox/s, {8.0 * n ** 3 / ms / 1e6:.1f} GB/s algorithmic, tile={tuple(info.last_tile)} '
      f'box={tuple(info.last_lds_dims)} lds={info.last_lds_bytes} prefilter_ms={info.prefilter_ms:.3f}')
