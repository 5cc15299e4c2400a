"""Do two HIP streams always give PCIe duplex?  Creates k extra streams between the upload and the download stream and
times 512 MiB up + 512 MiB down enqueued together (duplex: ~11 ms, serialised: ~18.7 ms)."""
import ctypes, time
import numpy as np
hip = ctypes.CDLL('libamdhip64.so')
def ck(e):
    if e != 0:
        raise RuntimeError(f'HIP error {e}')
vp = ctypes.c_void_p
N = 512 << 20
h_in = np.ones(N, dtype=np.uint8); h_out = np.zeros(N, dtype=np.uint8); h_out[:] = 2
ck(hip.hipHostRegister(vp(h_in.ctypes.data), ctypes.c_size_t(N), 0)); ck(hip.hipHostRegister(vp(h_out.ctypes.data), ctypes.c_size_t(N), 0))
d_a, d_b = vp(), vp()
ck(hip.hipMalloc(ctypes.byref(d_a), ctypes.c_size_t(N))); ck(hip.hipMalloc(ctypes.byref(d_b), ctypes.c_size_t(N)))
def stream():
    s = vp(); ck(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)); return s
def timed(s1, s2):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ck(hip.hipMemcpyAsync(d_a, vp(h_in.ctypes.data), ctypes.c_size_t(N), 1, s1))
        ck(hip.hipMemcpyAsync(vp(h_out.ctypes.data), d_b, ctypes.c_size_t(N), 2, s2))
        ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
streams = [stream() for _ in range(12)]
timed(streams[0], streams[1])
for j in range(1, 12):
    print(f'streams #0 and #{j}: {timed(streams[0], streams[j]):.2f} ms')
# a kernel-free dependency: download stream waits on an event of a third stream
