"""GPU time per transform (HIP events) and wall time per call at small/mid sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voltools_amd as vt
FLAGS = int(os.environ.get('MID_FLAGS', '0'))
SIZES = [int(x) for x in os.environ.get('MID_SIZES', '64,100,128,200,250,256,384').split(',')]
for n in SIZES:
    for interp in ('linear', 'filt_bspline'):
        vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        out = vt.empty((n, n, n), device='gpu:0')
        c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
        for label, ms in (('axis0', [vt.utils.transform_matrix(rotation=(0, i, 0), center=c) for i in range(0, 180, 3)]),
                          ('general', [vt.utils.transform_matrix(rotation=(i, 40 + i, 70 - i), rotation_order='sxyz', center=c) for i in range(0, 180, 3)])):
            for m in ms[:3]:
                sv.affine(m, output=out, _flags=FLAGS)
            sv.synchronize()
            t0 = time.perf_counter()
            sv.timer_start()
            for m in ms:
                sv.affine(m, output=out, _flags=FLAGS)
            gpu = sv.timer_stop() / len(ms) * 1e3
            wall = (time.perf_counter() - t0) / len(ms) * 1e6
            info = sv.info()
            print(f'{interp:13s} {n:4d}^3 {label:8s}: gpu {gpu:7.1f} us  wall {wall:7.1f} us/call  kernel={info.last_kernel} grid={info.last_grid} '
                  f'{n ** 3 / gpu / 1e3:.1f} Gvox/s')
        out.free(); sv.close()
