"""Where a budgeted config #4 sweep loses time: per sweep, the events' total, the copies built / evicted and their GPU time.
usage: [VT_MAX_RESIDENT_GB=8.5] [VT_DEBUG_ALLOC=1] python3 tools/diag/budget_sweep.py [size] [interp] [sweeps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
interp = sys.argv[2] if len(sys.argv) > 2 else 'filt_bspline'
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
out = vt.empty((n, n, n), device='gpu:0')
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
mats = [vt.utils.transform_matrix(rotation=(0, float(i), 0), center=c) for i in range(180)]
sv.affine(mats[0], output=out)
sv.synchronize()
prev = sv.info()
for s in range(sweeps):
    t0 = time.perf_counter()
    sv.timer_start()
    for m in mats:
        sv.affine(m, output=out)
    ms = sv.timer_stop()
    wall = (time.perf_counter() - t0) * 1e3
    i = sv.info()
    print('sweep %d: events %.2f ms (%.4f / step), wall %.2f ms; copies built %d (%.2f ms GPU), evicted %d; resident %.2f GiB, budget %.2f GiB'
          % (s, ms, ms / 180, wall, i.copies_built - prev.copies_built, i.copies_ms - prev.copies_ms, i.copies_evicted - prev.copies_evicted,
             i.resident_bytes / 2 ** 30, i.max_resident_bytes / 2 ** 30), flush=True)
    prev = i
