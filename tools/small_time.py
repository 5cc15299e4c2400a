"""Per-call cost at small sizes (launch-latency regime; README.md:74 of the reference quotes a 55 us floor at 5^3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voltools_amd as vt
for n in (5, 32, 64, 100, 128):
    for interp in ('linear', 'filt_bspline'):
        vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        out = vt.empty((n, n, n), device='gpu:0')
        c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
        ms = [vt.utils.transform_matrix(rotation=(0, i, 0), center=c) for i in range(180)]
        for m in ms[:5]:
            sv.affine(m, output=out)
        sv.synchronize()
        t0 = time.perf_counter()
        for m in ms:
            sv.affine(m, output=out)
        sv.synchronize()
        wall = (time.perf_counter() - t0) / len(ms) * 1e6
        line = f'{interp:13s} {n:4d}^3: {wall:7.1f} us/call (180 calls, python loop)'
        if hasattr(sv, 'affine_batch'):
            outs = vt.empty((len(ms), n, n, n), device='gpu:0')
            mm = np.stack(ms)
            sv.affine_batch(mm, output=outs)
            sv.synchronize()
            t0 = time.perf_counter()
            sv.affine_batch(mm, output=outs)
            sv.synchronize()
            wall_b = (time.perf_counter() - t0) / len(ms) * 1e6
            line += f'   batch: {wall_b:7.1f} us/matrix'
            outs.free()
        print(line)
        out.free(); sv.close()
