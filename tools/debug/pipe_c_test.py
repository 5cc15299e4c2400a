"""vt_affine_oneshot called directly (ctypes) with differently allocated host buffers: fresh numpy (registered inside the call),
numpy registered beforehand (what the result pool hands out), hipHostMalloc memory."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voltools_amd as vt
from voltools_amd import _native
lib = _native.load()
import importlib.util
_spec = importlib.util.find_spec('torch')
hip = ctypes.CDLL(os.path.join(list(_spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so'))   # the runtime the package bound to
n = 512
N = n ** 3
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
m = np.ascontiguousarray(vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=np.divide((n, n, n), 2)), dtype=np.float32)
fp = ctypes.POINTER(ctypes.c_float)
lib.vt_affine_oneshot.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
def run(tag, inp_ptr, out_ptr, reps=4):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = lib.vt_affine_oneshot(0, inp_ptr, n, n, n, 0, m.ctypes.data, out_ptr, 0, None)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, rc
    print(tag, ' '.join(f'{t:.2f}' for t in ts), 'ms', flush=True)
out_a = np.empty((n, n, n), np.float32); out_a[:] = 0
run('fresh numpy out (registered per call)   :', data.ctypes.data, out_a.ctypes.data)
out_b = np.empty((n, n, n), np.float32); out_b[:] = 0
assert hip.hipHostRegister(ctypes.c_void_p(out_b.ctypes.data), ctypes.c_size_t(N * 4), 0) == 0
run('numpy out registered beforehand          :', data.ctypes.data, out_b.ctypes.data)
assert hip.hipHostRegister(ctypes.c_void_p(data.ctypes.data), ctypes.c_size_t(N * 4), 0) == 0
run('both registered beforehand               :', data.ctypes.data, out_b.ctypes.data)
pin_in, pin_out = ctypes.c_void_p(), ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(pin_in), ctypes.c_size_t(N * 4), 0) == 0
assert hip.hipHostMalloc(ctypes.byref(pin_out), ctypes.c_size_t(N * 4), 0) == 0
ctypes.memmove(pin_in, data.ctypes.data, N * 4)
run('hipHostMalloc in and out                 :', pin_in.value, pin_out.value)
res = np.ctypeslib.as_array(ctypes.cast(pin_out, fp), shape=(N,)).reshape(n, n, n)
print('hipHostMalloc result equals numpy result:', bool(np.array_equal(res, out_b)))
