"""Prefilter timing per create: python3 tools/prefilter_time.py [size]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
d = vt.DeviceArray.from_numpy(vol, 0)
for i in range(3):
    sv = vt.StaticVolume(d, interpolation='filt_bspline', device='gpu:0')
    print(n, 'prefilter_ms', round(sv.info().prefilter_ms, 3), 'GB/s', round(24.0 * n ** 3 / sv.info().prefilter_ms / 1e6, 1))
    sv.close()
