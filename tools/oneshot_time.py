"""One-shot transform() (numpy in, numpy out) and StaticVolume -> numpy at 250^3 / 512^3, for a general rotation (plain
upload / transform / download sequence) and a rotation about axis 0 (slab pipeline, duplex PCIe)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt
for n in (250, 512):
    data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
    c = np.divide((n, n, n), 2)
    for mname, m in (('general', vt.utils.transform_matrix(rotation=(10, 20, 30), rotation_order='sxyz', center=c)),
                     ('axis0', vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=c))):
        for interp in ('linear', 'filt_bspline'):
            for _ in range(3):                    # warm-up while holding a result: the pool then owns both buffers the loop alternates between
                r = vt.affine(data, m, interpolation=interp, device='gpu')
            t0 = time.perf_counter()
            for _ in range(5):
                r = vt.affine(data, m, interpolation=interp, device='gpu')
            dt = (time.perf_counter() - t0) / 5
            sv = vt.StaticVolume(data, interpolation=interp, device='gpu')
            sv.affine(m)
            t0 = time.perf_counter()
            for _ in range(5):
                sv.affine(m)
            dt2 = (time.perf_counter() - t0) / 5
            sv.close()
            print(os.environ.get('VT_ONESHOT_SEQ', 'pipe'), n, mname, interp, f'transform() numpy in/out {dt*1e3:.2f} ms ; '
                  f'StaticVolume->numpy {dt2*1e3:.2f} ms ({n**3*4/dt2/1e9:.1f} GB/s D2H)')
