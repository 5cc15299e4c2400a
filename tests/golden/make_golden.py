"""Generate the golden fixtures under tests/golden/ from the reference's importable CPU path.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

What is recorded (inputs are re-creatable from the stored seeds/parameters; outputs are the data):
  matrices.npz ....... outputs of the reference's voltools.utils matrix builders
                       (/root/reference/voltools/utils/matrices.py:22-154) for a parameter grid that
                       covers all 24 rotation orders, both units, centre on/off, partial kwargs.
  volumes_m12.npz .... a seeded 48x52x56 float32 volume (large enough to have voxels 12 samples from every
                       face) and the reference CPU path's filt_bspline outputs for three matrices: pins the
                       prefiltered interpolations at the SURVEY 8c tolerance, 2e-6 at margin 12.
  launch_dims.npz .... outputs of the reference's compute_prefilter_workgroup_dims (utils/general.py:9-33; a pure function)
                       for a list of shapes: pins the informational helper of the same name.
  volumes.npz ........ a seeded 20x24x28 float32 volume and the outputs of
                       voltools.affine / voltools.transform / StaticVolume(device='cpu')
                       (/root/reference/voltools/transforms.py:109-162, volume.py:93-101) for
                       linear / bspline / bspline_simple / filt_bspline and several matrices,
                       plus reshape=True output shapes and padding
                       (/root/reference/voltools/utils/general.py:92-123).
"""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

with redirect_stdout(io.StringIO()):      # the reference prints a cupy warning at import
    import voltools as vt

assert vt.__file__.startswith('/root/reference'), vt.__file__


def golden_matrices():
    rs = np.random.RandomState(20261003)
    params, mats = [], []
    orders = vt.utils.AVAILABLE_ROTATIONS
    for i, order in enumerate(orders):
        for units in ('deg', 'rad'):
            ang = rs.uniform(-180, 180, 3) if units == 'deg' else rs.uniform(-np.pi, np.pi, 3)
            params.append(('rotation', order, units, *ang, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
            mats.append(vt.utils.rotation_matrix(tuple(ang), units, order))
    # the survey's probe: rotation=(0,45,0) rzxz about (N-1)/2 for N=200
    c = np.divide(np.subtract((200, 200, 200), 1), 2, dtype=np.float32)
    params.append(('probe200', 'rzxz', 'deg', 0, 45, 0, *c, 0, 0, 0, 0, 0, 0, 0, 0, 0))
    mats.append(vt.utils.transform_matrix(rotation=(0, 45, 0), center=c))
    # full compositions
    for i in range(48):
        order = orders[i % 24]
        ang = rs.uniform(-180, 180, 3)
        cen = rs.uniform(0, 300, 3).astype(np.float32)
        tr = rs.uniform(-20, 20, 3)
        sc = rs.uniform(0.5, 1.8, 3)
        sh = rs.uniform(-0.3, 0.3, 3)
        use = rs.rand(5) < 0.7           # rotation, center, translation, scale, shear
        kw = dict(rotation=tuple(ang) if use[0] else None, rotation_order=order,
                  center=tuple(cen) if use[1] else None, translation=tuple(tr) if use[2] else None,
                  scale=tuple(sc) if use[3] else None, shear=tuple(sh) if use[4] else None)
        params.append(('compose:' + ''.join('1' if u else '0' for u in use), order, 'deg', *ang, *cen, *tr, *sc, *sh))
        mats.append(vt.utils.transform_matrix(**kw))
    for t in [(1.5, -2.25, 3.0)]:
        params.append(('translation', '', '', 0, 0, 0, 0, 0, 0, *t, 0, 0, 0, 0, 0, 0))
        mats.append(vt.utils.translation_matrix(t))
        params.append(('scale', '', '', 0, 0, 0, 0, 0, 0, 0, 0, 0, *t, 0, 0, 0))
        mats.append(vt.utils.scale_matrix(t))
        params.append(('shear', '', '', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, *t))
        mats.append(vt.utils.shear_matrix(t))
    np.savez_compressed(os.path.join(HERE, 'matrices.npz'),
                        kind=np.array([p[0] for p in params]), order=np.array([p[1] for p in params]),
                        units=np.array([p[2] for p in params]),
                        values=np.array([p[3:] for p in params], dtype=np.float64),
                        matrices=np.stack(mats).astype(np.float32))
    print('matrices:', len(mats))


def golden_volumes():
    shape = (20, 24, 28)
    seed = 7
    vol = np.random.RandomState(seed).random_sample(shape).astype(np.float32)
    out = {'shape': np.array(shape), 'seed': np.array(seed)}

    center = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    cases = {
        'rot_inplane': dict(rotation=(0, 45, 0), rotation_order='rzxz'),
        'rot_general': dict(rotation=(25.0, -40.0, 70.0), rotation_order='sxyz'),
        'rot_scale_shift': dict(rotation=(10.0, 20.0, 30.0), rotation_order='rzxz', scale=(1.1, 0.9, 1.25),
                                translation=(1.5, -2.0, 0.75)),
        'shear': dict(shear=(0.1, -0.05, 0.2)),
    }
    interps = ['linear', 'bspline', 'bspline_simple', 'filt_bspline', 'filt_bspline_simple']
    for name, kw in cases.items():
        m = vt.utils.transform_matrix(center=center, **kw)
        out[f'{name}/matrix'] = m
        for interp in interps:
            with redirect_stdout(io.StringIO()):
                out[f'{name}/{interp}'] = vt.affine(vol, m, interpolation=interp, device='cpu')
    # transform() front end with the default centre and a float scale
    out['frontend/transform'] = vt.transform(vol, rotation=(0, 30, 0), scale=1.2, interpolation='filt_bspline', device='cpu')
    out['frontend/rotate'] = vt.rotate(vol, (15, 25, 35), rotation_order='szyx', interpolation='linear', device='cpu')
    out['frontend/translate'] = vt.translate(vol, (2, -1, 3), interpolation='linear', device='cpu')
    out['frontend/scale'] = vt.scale(vol, 1.5, interpolation='bspline', device='cpu')
    out['frontend/shear'] = vt.shear(vol, 0.1, interpolation='linear', device='cpu')
    # StaticVolume on the CPU device
    sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='cpu')
    out['static/transform'] = sv.transform(rotation=(0, 60, 0), translation=(1, 2, 3))
    # output= on the CPU path is returned and written
    buf = np.full(shape, 7.0, dtype=np.float32)
    res = vt.affine(vol, out['rot_general/matrix'], interpolation='linear', output=buf, device='cpu')
    assert res is buf
    out['frontend/output_arg'] = buf
    # reshape=True
    m = vt.utils.transform_matrix(rotation=(0, 30, 0), center=center)
    pb, pa, nd = vt.utils.compute_post_transform_dimensions(shape, m)
    out['reshape/matrix'] = m
    out['reshape/pad_before'], out['reshape/pad_after'], out['reshape/new_dims'] = pb, pa, nd
    out['reshape/linear'] = vt.affine(vol, m, interpolation='linear', reshape=True, device='cpu')
    np.savez_compressed(os.path.join(HERE, 'volumes.npz'), **out)
    print('volumes:', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'volumes.npz')) // 1024, 'KiB')


def golden_volumes_m12():
    """48x52x56: the smallest comfortable volume with an interior 12 samples from every face (|z|^12 = 1.4e-7 of boundary
    influence, the reference's own horizon, bspline.h:7), so filt_* is pinned at 2e-6 instead of 3e-5 at margin 8."""
    shape = (48, 52, 56)
    seed = 11
    vol = np.random.RandomState(seed).random_sample(shape).astype(np.float32)
    out = {'shape': np.array(shape), 'seed': np.array(seed)}
    center = np.divide(np.subtract(shape, 1), 2, dtype=np.float32)
    cases = {
        'rot_inplane': dict(rotation=(0, 45, 0), rotation_order='rzxz'),
        'rot_general': dict(rotation=(25.0, -40.0, 70.0), rotation_order='sxyz'),
        'rot_scale_shift': dict(rotation=(10.0, 20.0, 30.0), rotation_order='rzxz', scale=(1.1, 0.9, 1.25),
                                translation=(1.5, -2.0, 0.75)),
    }
    for name, kw in cases.items():
        m = vt.utils.transform_matrix(center=center, **kw)
        out[f'{name}/matrix'] = m
        with redirect_stdout(io.StringIO()):
            a = vt.affine(vol, m, interpolation='filt_bspline', device='cpu')
            b = vt.affine(vol, m, interpolation='filt_bspline_simple', device='cpu')
        assert np.array_equal(a, b)          # one scipy call serves both names on the CPU path (transforms.py:126-134)
        out[f'{name}/filt_bspline'] = a
    np.savez_compressed(os.path.join(HERE, 'volumes_m12.npz'), **out)
    print('volumes_m12:', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'volumes_m12.npz')) // 1024, 'KiB')


def golden_launch_dims():
    shapes = [(64, 64, 64), (200, 200, 200), (512, 512, 512), (48, 96, 40), (1, 5, 7), (30, 20, 12), (1024, 1024, 1024),
              (96, 64, 160), (250, 250, 250), (2, 3, 130), (128, 4, 36), (20, 24, 28)]
    grids, blocks = [], []
    for s in shapes:
        g, b = vt.utils.compute_prefilter_workgroup_dims(s)
        grids.append(g); blocks.append(b)
    np.savez_compressed(os.path.join(HERE, 'launch_dims.npz'), shapes=np.array(shapes), grids=np.array(grids), blocks=np.array(blocks))
    print('launch_dims:', len(shapes))


if __name__ == '__main__':
    which = sys.argv[1:] or ['matrices', 'volumes', 'volumes_m12', 'launch_dims']
    if 'launch_dims' in which:
        golden_launch_dims()
    if 'matrices' in which:
        golden_matrices()
    if 'volumes' in which:
        golden_volumes()
    if 'volumes_m12' in which:
        golden_volumes_m12()
