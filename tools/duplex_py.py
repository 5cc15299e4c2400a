"""PCIe duplex from inside a Python process (ctypes -> libamdhip64), numpy buffers registered in place: the same sequence as
tools/probes/duplex_probe.hip.  Separates "Python process / registered numpy memory" from "the library's pipeline code".
    python3 tools/duplex_py.py [--torch]     (--torch imports torch first, as the package's users do)"""
import ctypes, sys, time
import numpy as np
if '--torch' in sys.argv:
    import torch  # noqa: F401
    torch.cuda.is_available()
import os
if '--vt' in sys.argv:      # let the package initialise first (its HIP runtime, its kernels, one transform), then run the raw sequence
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import importlib.util
    import voltools_amd as vt
    _v = np.random.RandomState(0).random_sample((64, 64, 64)).astype(np.float32)
    vt.affine(_v, np.eye(4, dtype=np.float32), device='gpu')
    if '--vtbig' in sys.argv:
        _b = np.zeros((256, 256, 256), np.float32)
        vt.affine(_b, np.eye(4, dtype=np.float32), device='gpu')
    _spec = importlib.util.find_spec('torch')
    hip = ctypes.CDLL(os.path.join(list(_spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so'))
else:
    hip = ctypes.CDLL('libamdhip64.so')
def ck(e, what=''):
    if e != 0:
        raise RuntimeError(f'HIP error {e} {what}')
vp = ctypes.c_void_p
N = 512 << 20
h_in = np.ones(N, dtype=np.uint8)
h_out = np.zeros(N, dtype=np.uint8)
h_out[:] = 2
ck(hip.hipHostRegister(vp(h_in.ctypes.data), ctypes.c_size_t(N), 0), 'register')
ck(hip.hipHostRegister(vp(h_out.ctypes.data), ctypes.c_size_t(N), 0), 'register')
d_a, d_b = vp(), vp()
ck(hip.hipMalloc(ctypes.byref(d_a), ctypes.c_size_t(N)))
ck(hip.hipMalloc(ctypes.byref(d_b), ctypes.c_size_t(N)))
s1, s2 = vp(), vp()
ck(hip.hipStreamCreateWithFlags(ctypes.byref(s1), 1))
ck(hip.hipStreamCreateWithFlags(ctypes.byref(s2), 1))
H2D, D2H = 1, 2
def cp(dst, src, n, kind, s):
    ck(hip.hipMemcpyAsync(vp(dst), vp(src), ctypes.c_size_t(n), kind, s))
for rep in range(2):
    t0 = time.perf_counter()
    cp(d_a.value, h_in.ctypes.data, N, H2D, s1); ck(hip.hipStreamSynchronize(s1))
    t1 = time.perf_counter()
    cp(h_out.ctypes.data, d_b.value, N, D2H, s1); ck(hip.hipStreamSynchronize(s1))
    t2 = time.perf_counter()
    print(f'sequential: H2D {(t1-t0)*1e3:.2f} ms  D2H {(t2-t1)*1e3:.2f} ms  total {(t2-t0)*1e3:.2f}')
    t0 = time.perf_counter()
    cp(d_a.value, h_in.ctypes.data, N, H2D, s1)
    cp(h_out.ctypes.data, d_b.value, N, D2H, s2)
    ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
    print(f'concurrent whole copies on two streams: {(time.perf_counter()-t0)*1e3:.2f} ms')
    nch = 16; C = N // nch
    ev = [vp() for _ in range(nch)]
    for e in ev:
        ck(hip.hipEventCreate(ctypes.byref(e)))
    t0 = time.perf_counter()
    def up(k):
        if k >= nch:
            return
        cp(d_a.value + k * C, h_in.ctypes.data + k * C, C, H2D, s1)
        ck(hip.hipEventRecord(ev[k], s1))
    up(0); up(1)
    for k in range(nch):
        ck(hip.hipEventSynchronize(ev[k]))
        cp(h_out.ctypes.data + k * C, d_b.value + k * C, C, D2H, s2)
        up(k + 2)
    ti = time.perf_counter() - t0
    ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
    print(f'progressive pipeline: issue {ti*1e3:.2f} ms, total {(time.perf_counter()-t0)*1e3:.2f} ms')
    # dependent version: the download of chunk k reads what the upload of chunk k wrote (same device buffer)
    t0 = time.perf_counter()
    up(0); up(1)
    for k in range(nch):
        ck(hip.hipEventSynchronize(ev[k]))
        cp(h_out.ctypes.data + k * C, d_a.value + k * C, C, D2H, s2)
        up(k + 2)
    ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
    print(f'progressive pipeline, download reads the uploaded chunk: total {(time.perf_counter()-t0)*1e3:.2f} ms')
    for e in ev:
        ck(hip.hipEventDestroy(e))
