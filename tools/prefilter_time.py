"""One-time prefilter cost of a resident filt_bspline volume (vt_volume_create's three passes), warm: python3 tools/prefilter_time.py [size ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [512]:
    if n >= 768:
        import torch
        vol = torch.rand((n, n, n), dtype=torch.float32, device='cuda:0')
    else:
        vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
    ms = []
    for _ in range(5):
        sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu:0')
        ms.append(float(sv.info().prefilter_ms))
        sv.close()
    warm = min(ms[1:])
    print(f'prefilter {n}^3: first {ms[0]:.3f} ms, warm {warm:.3f} ms = {24.0 * n ** 3 / warm / 1e9:.2f} TB/s algorithmic (24 B/voxel), all: ' +
          ' '.join(f'{m:.3f}' for m in ms), flush=True)
