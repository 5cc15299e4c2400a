"""Replays tests/test_gpu_fuzz.py seed by seed until a case fails and prints where: plan, error map summary.   python3 tools/diag/span_case.py [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle
import test_gpu_fuzz as tf

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rs = np.random.RandomState(1000 + seed)
dims = [1, 2, 3, 5, 8, 17, 31, 33, 48, 64, 65, 97, 130, 200]
for _ in range(6):
    shape = tuple(int(rs.choice(dims)) for _ in range(3))
    if np.prod(shape) > 1.5e6:
        shape = (shape[0], min(shape[1], 64), shape[2])
    vol = rs.random_sample(shape).astype(np.float32)
    interp = rs.choice(list(tf.TOL))
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for kind in rs.choice(tf.KINDS, 3, replace=False):
        m = tf.random_matrix(rs, shape, kind)
        want = oracle.affine(vol, m, interp)
        for flags in rs.choice(len(tf.FLAG_SETS), 3, replace=False):
            fl = int(tf.FLAG_SETS[flags])
            got = sv.affine(m, _flags=fl)
            err = np.abs(got - want)
            info = sv.info()
            if err.max() > tf.TOL[interp]:
                print('FAIL', shape, interp, kind, fl, 'kernel', info.last_kernel, 'tile', tuple(info.last_tile), 'box', tuple(info.last_lds_dims), 'lds', info.last_lds_bytes, 'grid', info.last_grid)
                print(np.asarray(m))
                bad = np.argwhere(err > tf.TOL[interp])
                print('bad voxels', len(bad), 'of', err.size, 'first', bad[:10].tolist(), 'last', bad[-5:].tolist())
                print('d range', bad[:, 0].min(), bad[:, 0].max(), 'h range', bad[:, 1].min(), bad[:, 1].max(), 'w range', bad[:, 2].min(), bad[:, 2].max())
                for b in bad[:6]:
                    print(tuple(b), 'got', got[tuple(b)], 'want', want[tuple(b)])
                direct = sv.affine(m, _flags=_native.FORCE_DIRECT)
                print('direct ok:', float(np.abs(direct - want).max()))
                sys.exit(0)
    sv.close()
print('no failure for seed', seed)
