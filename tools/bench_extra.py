"""Secondary measurements for the BASELINE configurations that are not the headline bench line.

  config #2  512^3 linear, StaticVolume resident, output= device buffer
  config #3  512^3 filt_bspline: one-time prefilter and steady-state transform reported separately
  config #4  1024^3 filt_bspline, 180-step sweep `rotate((0, i, 0))` (README.md:25-27): total and per step
  protocol   the reference's own benchmark (tests/benchmark.py): 250^3, 100 random `sxyz` rotations about size/2,
             methods scipy / transform() numpy in+out / StaticVolume / StaticVolume + output=  (README.md:66-100)
Writes one JSON object to stdout.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt  # noqa: E402


def sweep(sv, out, mats):
    sv.affine(mats[0], output=out)
    sv.synchronize()
    sv.timer_start()
    for m in mats:
        sv.affine(m, output=out)
    return sv.timer_stop()


def resident(n, interp, nsteps):
    vol = np.random.RandomState(0).random_sample((n, n, n)).astype(np.float32)
    t0 = time.perf_counter()
    sv = vt.StaticVolume(vol, interpolation=interp, device='gpu:0')
    create_s = time.perf_counter() - t0
    out = vt.empty((n, n, n), device='gpu:0')
    c = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
    mats = [vt.utils.transform_matrix(rotation=(0, float(i), 0), center=c) for i in range(nsteps)]
    # The FIRST sweep of a fresh handle also builds the lazily created resident copies (plane-quad, and the in-plane transposed copy
    # from 45 degrees on): that is the honest figure for the README loop run once (`/root/reference/README.md:25-27`); the second sweep is
    # the steady state bench.py's headline line measures.  Both are printed.
    cold_ms = sweep(sv, out, mats)
    total_ms = sweep(sv, out, mats)
    info = sv.info()
    res = {'size': n, 'interpolation': interp, 'steps': nsteps, 'total_ms': round(total_ms, 3),
           'ms_per_step': round(total_ms / nsteps, 4), 'cold_first_sweep_total_ms': round(cold_ms, 3),
           'cold_first_sweep_ms_per_step': round(cold_ms / nsteps, 4), 'Mvoxels_per_s': round(n ** 3 * nsteps / total_ms / 1e3, 1),
           'algorithmic_GBps': round(8.0 * n ** 3 * nsteps / total_ms / 1e6, 1), 'prefilter_ms_once': round(float(info.prefilter_ms), 3),
           'create_wall_s_incl_upload': round(create_s, 3), 'kernel': int(info.last_kernel), 'resident_GiB': round(info.resident_bytes / 2 ** 30, 2)}
    sv.close()
    out.free()
    return res


def reference_protocol(n=250, nrot=100):
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
        # StaticVolume + output= (README 'static_vol_affine_out'): HIP events
        sv.affine(mats[0], output=out)
        sv.synchronize()
        sv.timer_start()
        for m in mats:
            sv.affine(m, output=out)
        row['static_vol_affine_out_ms'] = round(sv.timer_stop() / nrot, 4)
        # StaticVolume returning numpy (README 'static_vol_affine'): wall clock, includes D2H
        t0 = time.perf_counter()
        for m in mats[:20]:
            sv.affine(m)
        row['static_vol_affine_ms'] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        # transform() numpy in / numpy out (README 'transforms_affine'): wall clock, includes H2D + prefilter + D2H
        t0 = time.perf_counter()
        for m in mats[:10]:
            vt.affine(data, m, interpolation=interp, device='gpu')
        row['transforms_affine_ms'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        # scipy (the reference CPU path), 2 rotations only
        from scipy.ndimage import affine_transform
        t0 = time.perf_counter()
        for m in mats[:2]:
            affine_transform(data, m, order=order)
        row['scipy_ms'] = round((time.perf_counter() - t0) / 2 * 1e3, 1)
        row['kernel'] = int(sv.info().last_kernel)
        res[interp] = row
        sv.close()
        out.free()
    return res


if __name__ == '__main__':
    result = {
        'config2_512_linear': resident(512, 'linear', 180),
        'config3_512_filt_bspline': resident(512, 'filt_bspline', 180),
        'config4_1024_filt_bspline_180_sweep': resident(1024, 'filt_bspline', 180),
        'config_1024_linear_180_sweep': resident(1024, 'linear', 180),
        'reference_protocol_250_100_random_sxyz': reference_protocol(),
    }
    print(json.dumps(result, indent=1))
