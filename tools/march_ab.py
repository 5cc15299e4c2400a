"""A/B timing of the axis-0 marching kernels over a sweep of README angles, in ONE process (interleaved rounds).

    python3 tools/march_ab.py --size 512 --interp linear filt_bspline --flags 0 2048 [--angles 0 180 5] [--rounds 3]

Prints per (interp, flags): mean / min / max ms per launch over the angles, algorithmic TB/s (8 B/voxel) and the fraction of
8 TB/s; kernel id and tile of the last launch.  Environment knobs (VT_TILE, VT_DCH, ...) are read at handle creation, so
`--env NAME=VALUE[,NAME=VALUE]` variants get their own handles.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--interp', nargs='+', default=['linear', 'filt_bspline'])
ap.add_argument('--flags', nargs='+', type=int, default=[0, 2048])
ap.add_argument('--env', nargs='*', default=[''], help='environment variants, e.g. VT_DCH=128 "VT_TILE=2,VT_DCH=64"')
ap.add_argument('--angles', nargs=3, type=float, default=[0, 180, 5])
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--per-angle', action='store_true', help='print the per-angle times of every variant')
args = ap.parse_args()

n = args.size
if n >= 1024:
    import torch
    g = torch.Generator(device='cuda:0'); g.manual_seed(0)
    vol = torch.rand((n, n, n), dtype=torch.float32, device='cuda:0', generator=g)
else:
    vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
out = vt.empty((n, n, n), device='gpu:0')
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
angles = np.arange(*args.angles)
mats = [vt.utils.transform_matrix(rotation=(0, float(a), 0), rotation_order='rzxz', center=c) for a in angles]

for interp in args.interp:
    variants = []
    for env in args.env:
        kv = dict(x.split('=') for x in env.split(',') if x)
        for k, v_ in kv.items():
            os.environ[k] = v_
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for k in kv:
            del os.environ[k]
        for fl in args.flags:
            variants.append((env, fl, sv))
    times = {(e, f): np.zeros(len(mats)) for e, f, _ in variants}
    kinfo = {}
    for e, f, sv in variants:                       # warm-up: builds the resident copies
        for m in mats[:2]:
            sv.affine(m, output=out, _flags=f)
        sv.synchronize()
    for r in range(args.rounds):
        for e, f, sv in variants:
            for i, m in enumerate(mats):
                sv.timer_start()
                for _ in range(args.reps):
                    sv.affine(m, output=out, _flags=f)
                ms = sv.timer_stop() / args.reps
                times[(e, f)][i] = ms if r == 0 else min(times[(e, f)][i], ms)
            info = sv.info()
            kinfo[(e, f)] = (info.last_kernel, tuple(info.last_tile), info.last_lds_bytes, info.last_grid)
    for e, f, sv in variants:
        t = times[(e, f)]
        tb = 8.0 * n ** 3 / (t.mean() * 1e-3) / 1e12
        k = kinfo[(e, f)]
        worst = angles[int(np.argmax(t))]
        print(f'{interp:14s} {n}^3 flags={f:5d} env={e or "-":24s} kernel={k[0]} tile={k[1]} lds={k[2]} grid={k[3]}: '
              f'mean {t.mean():.4f} ms  min {t.min():.4f}  max {t.max():.4f} (at {worst:.0f} deg)  {tb:.2f} TB/s = {tb / 8 * 100:.1f} %', flush=True)
        if args.per_angle:
            print('    ' + ' '.join(f'{a:.0f}:{x * 1e3:.0f}' for a, x in zip(angles, t)), flush=True)
    seen = set()
    for e, f, sv in variants:
        if id(sv) not in seen:
            seen.add(id(sv)); sv.close()
