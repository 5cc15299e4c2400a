"""Would general rotations run faster on a resident copy whose FASTEST axis is the source axis the output's w direction follows most closely?
Emulated from Python: three handles hold the volume with its source axes permuted ((0,1,2) plain, (0,2,1) rows along y, (2,1,0) rows along z),
the pull matrix's rows are permuted to match (same output), and every one of the reference protocol's 100 random rotations is timed on the
plain handle and on the handle its w column asks for.    python3 tools/general_reorient_probe.py [size] [interp ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voltools_amd as vt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
interps = sys.argv[2:] or ['linear', 'filt_bspline']
rs = np.random.RandomState(1)
data = rs.random_sample((n, n, n)).astype(np.float32)
rots = rs.uniform(-180, 180, (100, 3))
mats = [np.asarray(vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)), np.float32) for r in rots]
perms = {2: (0, 1, 2), 1: (0, 2, 1), 0: (2, 1, 0)}            # fastest source axis -> axis order of the copy
out = vt.zeros((n, n, n), device='gpu:0')
for interp in interps:
    hs = {a: vt.StaticVolume(np.ascontiguousarray(data.transpose(p)), interpolation=interp, device='gpu:0') for a, p in perms.items()}
    def permuted(m, p):
        mp = m.copy()
        mp[:3] = m[list(p)]
        return mp
    def timed(h, m, reps=6):
        h.affine(m, output=out)
        h.timer_start()
        for _ in range(reps):
            h.affine(m, output=out)
        return h.timer_stop() / reps
    # equality of the two routes on one matrix
    a = hs[2].affine(mats[0]); b = hs[0].affine(permuted(mats[0], perms[0])); c = hs[1].affine(permuted(mats[0], perms[1]))
    print(f'{interp}: max |plain - z-rows| {np.abs(a - b).max():.2e}, |plain - y-rows| {np.abs(a - c).max():.2e}', flush=True)
    base, best, allo = [], [], []
    for m in mats:
        t = {a: timed(hs[a], permuted(m, perms[a])) for a in perms}
        want = int(np.argmax(np.abs(m[:3, 2])))
        base.append(t[2]); best.append(t[want]); allo.append(min(t.values()))
    print(f'{n}^3 {interp}: plain copy {np.mean(base):.4f} ms | copy chosen by the w column {np.mean(best):.4f} ms | best of the three per matrix {np.mean(allo):.4f} ms '
          f'(kernel {hs[2].info().last_kernel})', flush=True)
    for h in hs.values():
        h.close()
