"""Where the one-shot transform() time goes (250^3 / 512^3, numpy in -> numpy out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voltools_amd as vt
def t(fn, reps=7):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0); del r
    return best * 1e3
for n in [int(x) for x in os.environ.get("ONESHOT_SIZES", "250,512").split(",")]:
    data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
    m = vt.utils.transform_matrix(rotation=(10, 20, 30), rotation_order='sxyz', center=np.divide((n, n, n), 2))
    for interp in ('linear', 'filt_bspline'):
        total = t(lambda: vt.affine(data, m, interpolation=interp, device='gpu'))
        def create():
            sv = vt.StaticVolume(data, interpolation=interp, device='gpu'); return sv
        svs = []
        c = t(lambda: svs.append(create()) or svs.pop().close())
        sv = create()
        a = t(lambda: sv.affine(m))
        out = vt.empty((n, n, n), device='gpu:0')
        k = t(lambda: (sv.affine(m, output=out), sv.synchronize()))
        print(f'{n}^3 {interp:13s}: transform() {total:6.2f} ms | create+destroy (H2D, alloc, prefilter) {c:6.2f} | affine -> numpy (kernel + D2H) {a:6.2f} | kernel only {k:5.2f}'
              f' | ideal PCIe 2 x {n**3*4/50e6:.2f} ms')
        sv.close(); out.free()
