"""Mean ms per transform over the 100 random `sxyz` rotations of the reference's protocol (tests/benchmark.py:52-54), one variant per
process (knobs come from the environment):   VT_BLOCK_NT=512 python3 tools/general_time.py [size] [interp ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
interps = sys.argv[2:] or ['filt_bspline']
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rs.uniform(-180, 180, (100, 3))]
for interp in interps:
    sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
    out = vt.zeros((n, n, n), device='gpu:0')
    row = []
    for _ in range(3):
        for m in mats[:5]:
            sv.affine(m, output=out)
        sv.synchronize()
        sv.timer_start()
        for m in mats:
            sv.affine(m, output=out)
        row.append(sv.timer_stop() / len(mats))
    knobs = ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('VT_'))
    print(f'{n}^3 {interp} [{knobs or "default"}]: {min(row):.4f} ms (rounds {" ".join(f"{x:.4f}" for x in row)}), kernel {sv.info().last_kernel}', flush=True)
    sv.close()
