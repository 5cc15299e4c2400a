"""The reference's own benchmark protocol (tests/benchmark.py: sizes :37, 100 random `sxyz` rotations about size/2 :52-54,
methods :59-65), all sizes of its README tables (README.md:66-100), on this GPU.  Prints JSON; the like-for-like rows
against BASELINE.md section 1.  scipy is timed on 2 rotations and only up to 100^3."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt

def protocol(n, nrot=100):
    rs = np.random.RandomState(1)
    data = rs.random_sample((n, n, n)).astype(np.float32)
    rotations = rs.uniform(-180, 180, (nrot, 3))
    center = np.divide((n, n, n), 2)
    mats = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=center) for r in rotations]
    res = {}
    for interp, order in (('linear', 1), ('filt_bspline', 3), ('filt_bspline_simple', 3)):
        sv = vt.StaticVolume(data, interpolation=interp, device='gpu:0')
        out = vt.zeros((n, n, n), device='gpu:0')
        row = {}
        for m in mats[:5]:
            sv.affine(m, output=out)
        sv.synchronize()
        t0 = time.perf_counter()
        for m in mats:
            sv.affine(m, output=out)
        sv.synchronize()
        row['static_vol_out_ms'] = round((time.perf_counter() - t0) / nrot * 1e3, 4)          # README 'static_vol_affine_out'
        t0 = time.perf_counter()
        for m in mats:
            sv.affine(m)
        row['static_vol_ms'] = round((time.perf_counter() - t0) / nrot * 1e3, 4)                 # README 'static_vol_affine'
        t0 = time.perf_counter()
        for m in mats:                                # all 100, first call included, as the reference averages them
            vt.affine(data, m, interpolation=interp, device='gpu')
        row['np_transform_ms'] = round((time.perf_counter() - t0) / nrot * 1e3, 4)               # README 'transforms_affine' numpy in/out
        t0 = time.perf_counter()
        for m in mats:                                # numpy in, device `output=` (tests/benchmark.py:62): upload + prefilter + kernel per call, no download
            vt.affine(data, m, interpolation=interp, device='gpu', output=out)
        sv.synchronize()
        row['np_transform_out_ms'] = round((time.perf_counter() - t0) / nrot * 1e3, 4)           # README 'transforms_affine_out'
        outs = vt.empty((nrot, n, n, n), device='gpu:0') if n <= 100 else None
        if outs is not None:
            mm = np.stack(mats)
            sv.affine_batch(mm, output=outs); sv.synchronize()
            t0 = time.perf_counter()
            sv.affine_batch(mm, output=outs); sv.synchronize()
            row['batch_ms_per_matrix'] = round((time.perf_counter() - t0) / nrot * 1e3, 5)     # no counterpart in the reference
            outs.free()
        if n <= 100:
            from scipy.ndimage import affine_transform
            t0 = time.perf_counter()
            for m in mats[:2]:
                affine_transform(data, m, order=order)
            row['scipy_ms'] = round((time.perf_counter() - t0) / 2 * 1e3, 2)
        row['kernel'] = int(sv.info().last_kernel)
        res[interp] = row
        sv.close(); out.free()
    return res

if __name__ == '__main__':
    print(json.dumps({str(n): protocol(n) for n in (5, 10, 25, 50, 100, 250)}, indent=1))
