import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle
import test_gpu_fuzz as F
np.set_printoptions(precision=9, suppress=False, linewidth=200)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rs = np.random.RandomState(1000 + seed)
dims = [1, 2, 3, 5, 8, 17, 31, 33, 48, 64, 65, 97, 130, 200]
for _ in range(6):
    shape = tuple(int(rs.choice(dims)) for _ in range(3))
    if np.prod(shape) > 1.5e6:
        shape = (shape[0], min(shape[1], 64), shape[2])
    vol = rs.random_sample(shape).astype(np.float32)
    interp = rs.choice(list(F.TOL))
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for kind in rs.choice(F.KINDS, 3, replace=False):
        m = F.random_matrix(rs, shape, kind)
        want = oracle.affine(vol, m, interp)
        fl = rs.choice(len(F.FLAG_SETS), 3, replace=False)
        for flags in F.FLAG_SETS:
            got = sv.affine(m, _flags=int(flags))
            err = float(np.abs(got - want).max())
            info = sv.info()
            if not err <= F.TOL[interp]:
                bad = np.argwhere(~(np.abs(got - want) <= F.TOL[interp]))
                print('FAIL', shape, interp, kind, 'flags', flags, 'kernel', info.last_kernel, 'tile', list(info.last_tile), 'lds', list(info.last_lds_dims),
                      info.last_lds_bytes, 'grid', info.last_grid, 'err', err, 'nbad', len(bad), 'first', bad[:3].tolist(), 'min', bad.min(0).tolist(), 'max', bad.max(0).tolist())
                print(np.asarray(m, np.float64))
    sv.close()
print('done')
