"""Fuzz of the plane-quad marching kernel (kernel 8) on what round 3 added: the one-plane trilinear kernel (integer axis-0
offsets), the service-group lane mapping, the prefix-sum row placement, tiles 4 / 5 (512 threads) and the fused X+Y prefilter
underneath `filt_*`.  Random ragged shapes x axis-0-separable matrices (rotation, in-plane scale and shear, integer and
fractional axis-0 shifts, mirrors) x every interpolation x a forced tile, whole volume against the oracle.  Tile and knob
are read at create, so every case is its own handle.
    VT_DEBUG_GUARD=1 python3 tools/fuzz_quad.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import voltools_amd as vt
from voltools_amd import _native
from oracle import oracle

TOL = {'linear': 1e-6, 'bspline': 1e-6, 'bspline_simple': 1e-6, 'filt_bspline': 3e-6, 'filt_bspline_simple': 3e-6}
KNOBS = [{}, {'VT_QUAD_PERM': '0'}, {'VT_QUAD_ROWS': '-1'}, {'VT_QUAD_ZID': '0'}, {'VT_ZID_DCH': '8'}, {'VT_DCH': '8'}]


def separable_matrix(rs, shape):
    """4x4 output->source matrix whose row 0 is (±1, 0, 0, integer or fractional shift) and whose in-plane block is a
    rotation times a mild scale / shear."""
    c = (np.asarray(shape, dtype=np.float64) - 1) / 2
    a = np.deg2rad(rs.choice([0.0, 90.0, 180.0, rs.uniform(-180, 180), rs.uniform(-8, 8), 45.0, 30.0]))
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    kind = rs.randint(4)
    if kind == 1:
        R = R @ np.diag(rs.uniform(0.6, 1.6, 2))
    elif kind == 2:
        R = R @ np.array([[1.0, rs.uniform(-0.3, 0.3)], [0.0, 1.0]])
    elif kind == 3:
        R = R @ np.diag(rs.choice([-1.0, 1.0], 2))
    m = np.eye(4)
    m[1:3, 1:3] = R
    zs = rs.choice([0.0, 1.0, -3.0, float(rs.randint(-shape[0], shape[0] + 1)), rs.uniform(-3, 3), 0.5])
    m[0, 0] = rs.choice([1.0, 1.0, 1.0, -1.0])
    sh = np.array([zs, rs.uniform(-5, 5), rs.uniform(-5, 5)]) if rs.rand() < 0.7 else np.array([zs, 0.0, 0.0])
    # rotate about the centre: x_src = L (x_out - c) + c + shift
    L = m[:3, :3]
    m[:3, 3] = c - L @ c + sh
    if m[0, 0] < 0:
        m[0, 3] = np.round(m[0, 3]) if zs == np.round(zs) else m[0, 3]
    return m.astype(np.float32)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    rs = np.random.RandomState(4242)
    dims0 = [1, 3, 4, 5, 8, 17, 33, 64, 70]
    dims = [9, 17, 31, 33, 48, 64, 65, 97, 130, 200, 257, 300]
    served = {}
    worst = {k: 0.0 for k in TOL}
    for it in range(ncases):
        shape = (int(rs.choice(dims0)), int(rs.choice(dims)), int(rs.choice(dims)))
        interp = list(TOL)[it % len(TOL)]
        tile = int(rs.choice([-1, 0, 1, 2, 3, 4, 5]))
        knob = KNOBS[rs.randint(len(KNOBS))]
        env = dict(knob)
        if tile >= 0:
            env['VT_TILE'] = str(tile)
        for k, v in env.items():
            os.environ[k] = v
        vol = rs.random_sample(shape).astype(np.float32)
        sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
        for _ in range(3):
            m = separable_matrix(rs, shape)
            want = oracle.affine(vol, m, interp)
            got = sv.affine(m, _flags=_native.FORCE_TILED)
            k = sv.info().last_kernel
            served[k] = served.get(k, 0) + 1
            assert np.isfinite(got).all(), (it, shape, interp, env, k, m.tolist())
            err = float(np.abs(got - want).max())
            assert err <= TOL[interp], (it, shape, interp, env, k, err, m.tolist())
            worst[interp] = max(worst[interp], err)
        sv.close()
        for k in env:
            del os.environ[k]
        if it % 40 == 39:
            print('case', it + 1, 'kernels', served, flush=True)
    print('ok:', ncases * 3, 'launches; kernels', served, '; worst error per interpolation', worst)


if __name__ == '__main__':
    main()
