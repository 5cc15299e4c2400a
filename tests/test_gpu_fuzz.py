"""-m gpu: seeded random shapes x matrix classes x kernel-forcing flags against the oracle (whole volume)."""
import numpy as np
import pytest

import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = {'linear': 1e-6, 'bspline': 1e-6, 'bspline_simple': 1e-6, 'filt_bspline': 3e-6, 'filt_bspline_simple': 3e-6}
FLAG_SETS = (0, _native.FORCE_TILED, _native.FORCE_TILED | _native.NO_RSWAP, _native.FORCE_TILED | _native.FORCE_XSWAP, _native.FORCE_TILED | _native.NO_ZPAIR, _native.FORCE_TILED | _native.NO_QUAD, _native.FORCE_TILED | _native.NO_QUAD | _native.FORCE_XSWAP,
             _native.FORCE_TILED | _native.NO_MARCH, _native.FORCE_TILED | _native.NO_ZSEP | _native.FORCE_PACKED,
             _native.FORCE_TILED | _native.NO_ZSEP | _native.NO_PACKED, _native.FORCE_DIRECT,
             _native.FORCE_TILED | _native.NO_ZSEP, _native.FORCE_TILED | _native.NO_BLOCK)


def random_matrix(rs, shape, kind):
    c = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    ang = rs.uniform(-180, 180, 3)
    tr = tuple(rs.uniform(-4, 4, 3))
    if kind == 'axis0':
        return vt.utils.transform_matrix(rotation=(ang[0], 0, 0), rotation_order='sxyz', translation=tr, center=c)
    if kind == 'axis1':
        return vt.utils.transform_matrix(rotation=(0, ang[1], 0), rotation_order='sxyz', translation=tr, center=c)
    if kind == 'axis2':
        return vt.utils.transform_matrix(rotation=(0, 0, ang[2]), rotation_order='sxyz', translation=tr, center=c)
    if kind == 'general':
        return vt.utils.transform_matrix(rotation=tuple(ang), rotation_order='rzxz', translation=tr, center=c)
    if kind == 'affine':
        return vt.utils.transform_matrix(rotation=tuple(ang), scale=tuple(rs.uniform(0.4, 2.5, 3)), shear=tuple(rs.uniform(-0.3, 0.3, 3)),
                                         translation=tr, center=c)
    if kind == 'quarter':            # axis permutations / quarter turns: exact integer coordinates
        q = rs.choice([0, 90, 180, 270], 3)
        return vt.utils.transform_matrix(rotation=tuple(float(x) for x in q), rotation_order='sxyz', center=c)
    if kind == 'singular':           # one source axis collapsed
        m = vt.utils.transform_matrix(rotation=tuple(ang), translation=tr, center=c)
        m[rs.randint(3), :3] = 0.0
        return m
    if kind == 'far':
        return vt.utils.translation_matrix(tuple(rs.choice([-1, 1], 3) * rs.uniform(0, 3, 3) * np.array(shape)))
    raise KeyError(kind)


KINDS = ('axis0', 'axis1', 'axis2', 'general', 'affine', 'quarter', 'singular', 'far')


@pytest.mark.parametrize('seed', range(12))
def test_random_cases_match_oracle(seed):
    rs = np.random.RandomState(1000 + seed)
    dims = [1, 2, 3, 5, 8, 17, 31, 33, 48, 64, 65, 97, 130, 200]
    for _ in range(6):
        shape = tuple(int(rs.choice(dims)) for _ in range(3))
        if np.prod(shape) > 1.5e6:
            shape = (shape[0], min(shape[1], 64), shape[2])
        vol = rs.random_sample(shape).astype(np.float32)
        interp = rs.choice(list(TOL))
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for kind in rs.choice(KINDS, 3, replace=False):
            m = random_matrix(rs, shape, kind)
            want = oracle.affine(vol, m, interp)
            for flags in rs.choice(len(FLAG_SETS), 3, replace=False):
                got = sv.affine(m, _flags=int(FLAG_SETS[flags]))
                err = float(np.abs(got - want).max())
                assert err <= TOL[interp], (seed, shape, interp, kind, int(FLAG_SETS[flags]), sv.info().last_kernel, err)
        sv.close()
