"""Identical handles in one process (same volume, same planner choices), created and used one after another: sweep mean per handle, again
after all exist, and for a handle re-created at a closed one's addresses.  [measured, profiles/r03_placement_probe.txt] handles differ by up to 5-6 %,
persistently -- which resident copy a launch reads matters, not only what it does -- so A/B variants that live in different handles
(tools/march_ab.py) are read with that spread in mind, and decisive comparisons use one process per variant on one box.
    python3 tools/placement_probe.py [size] [interpolation]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
interp = sys.argv[2] if len(sys.argv) > 2 else 'filt_bspline'
vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
mats = [vt.utils.transform_matrix(rotation=(0, float(a), 0), rotation_order='rzxz', center=c) for a in range(0, 45, 3)]
out = vt.empty((n, n, n), device='gpu:0')
def sweep(sv):
    for m in mats[:3]:
        sv.affine(m, output=out)
    sv.synchronize()
    best = 1e9
    for _ in range(3):
        sv.timer_start()
        for m in mats:
            sv.affine(m, output=out)
        best = min(best, sv.timer_stop() / len(mats))
    return best
hs = []
for i in range(4):
    hs.append(vt.StaticVolume(vol, interpolation=interp, device='gpu:0'))
    print(f'handle {i}: {sweep(hs[-1]):.4f} ms', flush=True)
print('again:', ' '.join(f'{sweep(h):.4f}' for h in hs), flush=True)
hs[0].close()
h = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
print(f'after closing handle 0, a new handle: {sweep(h):.4f} ms; the others again:', ' '.join(f'{sweep(x):.4f}' for x in hs[1:]), flush=True)
