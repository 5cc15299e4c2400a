import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
for n in (512, 1024):
    vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
    ts = []
    for _ in range(5):
        sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu:0')
        ts.append(sv.info().prefilter_ms); sv.close()
    print(os.path.basename(os.path.dirname(os.environ.get('VT_LIB', 'lib/x'))), n, ' '.join(f'{t:.3f}' for t in ts), 'ms')
