"""Plan and time of the trilinear general-matrix kernel (kind 6) per matrix class: tile, LDS bytes, grid, ms.
python3 tools/span_probe.py [size]          (VT_LIB / VT_EXP_* select the ablation build and its switches)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
rots = rs.uniform(-180, 180, (100, 3))
mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)) for r in rots]
sv = vt.StaticVolume(data, interpolation='linear', device='gpu:0')
out = vt.zeros((n, n, n), device='gpu:0')
for _ in range(2):
    for m in mats[:8]:
        sv.affine(m, output=out)
sv.synchronize()
tot = 0.0
for i, m in enumerate(mats[:24]):
    sv.affine(m, output=out)
    sv.timer_start()
    for _ in range(8):
        sv.affine(m, output=out)
    ms = sv.timer_stop() / 8
    tot += ms
    info = sv.info()
    print(i, f'{ms:.4f} ms kernel {info.last_kernel} tile {tuple(info.last_tile)} lds {info.last_lds_bytes} grid {info.last_grid} box {tuple(info.last_lds_dims)}', flush=True)
if os.environ.get('VT_EXP_STAMPS'):
    # one more launch, then the phase timers the kernel wrote over the start of the output (100 MHz ticks of s_memtime)
    sv.affine(mats[1], output=out)
    sv.synchronize()
    g = int(sv.info().last_grid)
    raw = out.get().reshape(-1)[:16 * g].reshape(g, 16).astype(np.float64)
    names = ['set-up', 'id+decode+publish barrier', 'geometry', 'staging issue', 'base+masks', 'wait+barrier', 'gather', 'outside tiles', 'TOTAL',
             'RIM staging issue', 'RIM base+masks', 'RIM wait+barrier', 'RIM gather', '# fast tiles', '# rim tiles', '# outside tiles']
    tot_ = raw[:, 8].mean()
    for k, nm in enumerate(names):
        print(f'  phase {k} {nm:28s} mean {raw[:, k].mean():10.0f} ticks  ({100 * raw[:, k].mean() / tot_:5.1f} %)   min {raw[:, k].min():.0f} max {raw[:, k].max():.0f}')
knobs = ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('VT_'))
print(f'mean of 24: {tot / 24:.4f} ms [{knobs or "default"}]')
