"""Host<->device copy rates for the buffers the Python API meets: fresh numpy, touched numpy, pinned (hipHostMalloc)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voltools_amd as vt
from voltools_amd import _native
lib = _native.load()
hip = ctypes.CDLL(None)            # the HIP runtime is already loaded by _native
n = 512
nbytes = n ** 3 * 4
d = vt.empty((n, n, n), device='gpu:0')
def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best
def d2h(arr): _native.check(lib.vt_memcpy_d2h(0, ctypes.c_void_p(arr.ctypes.data), ctypes.c_void_p(d.ptr), ctypes.c_size_t(nbytes)), 'd2h')
def h2d(arr): _native.check(lib.vt_memcpy_h2d(0, ctypes.c_void_p(d.ptr), ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(nbytes)), 'h2d')
fresh = lambda: d2h(np.empty((n, n, n), np.float32))
print(f'D2H into fresh np.empty      : {t(fresh)*1e3:7.1f} ms')
a = np.zeros((n, n, n), np.float32); a += 1
print(f'D2H into touched numpy       : {t(lambda: d2h(a))*1e3:7.1f} ms')
print(f'H2D from touched numpy       : {t(lambda: h2d(a))*1e3:7.1f} ms')
os.environ['VT_NO_PIN'] = '1'
p = ctypes.c_void_p()
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
t0 = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(p), nbytes, 0); t1 = time.perf_counter()
print(f'hipHostMalloc 512 MiB        : {(t1-t0)*1e3:7.1f} ms rc={rc}')
buf = (ctypes.c_float * (n ** 3)).from_address(p.value)
pin = np.frombuffer(buf, dtype=np.float32).reshape(n, n, n)
print(f'D2H into pinned              : {t(lambda: d2h(pin))*1e3:7.1f} ms')
print(f'H2D from pinned              : {t(lambda: h2d(pin))*1e3:7.1f} ms')
b = np.empty((n, n, n), np.float32)
print(f'memcpy pinned -> fresh numpy : {t(lambda: np.copyto(np.empty((n, n, n), np.float32), pin))*1e3:7.1f} ms')
print(f'memcpy pinned -> touched     : {t(lambda: np.copyto(a, pin))*1e3:7.1f} ms')
print(f'np.empty + touch (fault in)  : {t(lambda: np.empty((n, n, n), np.float32).fill(0))*1e3:7.1f} ms')
