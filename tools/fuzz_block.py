"""Extended fuzz of the lane-block kernel (kernel 9): 60 random shapes x 6 matrix classes x 4 interpolations (trilinear through
VT_BLOCK_LINEAR), whole volume against the oracle; run it under VT_DEBUG_GUARD=1 so that an out-of-bounds read shows as NaN.
    VT_DEBUG_GUARD=1 python3 tools/fuzz_block.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle
from test_gpu_fuzz import random_matrix
rs = np.random.RandomState(77)
dims = [17, 31, 33, 48, 64, 65, 97, 130, 200]
n9 = 0; worst = 0
for it in range(60):
    shape = tuple(int(x) for x in rs.choice(dims, 3))
    vol = rs.random_sample(shape).astype(np.float32)
    interp = ['bspline', 'filt_bspline', 'bspline_simple', 'linear'][it % 4]
    os.environ['VT_BLOCK_LINEAR'] = '1'
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    for kind in ('general', 'affine', 'axis1', 'quarter', 'far', 'singular'):
        m = random_matrix(rs, shape, kind)
        want = oracle.affine(vol, m, interp)
        got = sv.affine(m, _flags=_native.FORCE_TILED | _native.NO_ZSEP)
        k = sv.info().last_kernel
        err = float(np.abs(got - want).max())
        assert np.isfinite(got).all(), (shape, interp, kind, k)
        tol = 1e-5 if interp.startswith('filt') else 2e-6
        assert err <= tol, (shape, interp, kind, k, err)
        n9 += k == 9; worst = max(worst, err)
    sv.close()
print('ok: kernel 9 served', n9, 'of', 60 * 6, 'cases; worst error', worst)
