"""Which per-call step of the one-shot breaks PCIe duplex?  The progressive pipeline of tools/duplex_py.py (512 MiB up + down,
16 chunks), one configuration per process:  python3 tools/duplex_fresh.py A|B|C|D|E|F
  A everything set up once          B device buffers from hipMalloc/hipFree on every repetition
  C input registered per repetition D output registered per repetition
  E both host buffers unregistered and registered again once, before the loop     F device buffers freed and allocated again once"""
import ctypes, sys, time
import numpy as np
cfg = sys.argv[1]
hip = ctypes.CDLL('libamdhip64.so')
def ck(e, what=''):
    if e != 0:
        raise RuntimeError(f'HIP error {e} {what}')
vp = ctypes.c_void_p
N = 512 << 20
h_in = np.ones(N, dtype=np.uint8)
h_out = np.zeros(N, dtype=np.uint8); h_out[:] = 2
s1, s2 = vp(), vp()
ck(hip.hipStreamCreateWithFlags(ctypes.byref(s1), 1)); ck(hip.hipStreamCreateWithFlags(ctypes.byref(s2), 1))
nch = 16; C = N // nch
ev = [vp() for _ in range(nch)]
for e in ev:
    ck(hip.hipEventCreate(ctypes.byref(e)))
def cp(dst, src, n, kind, s):
    ck(hip.hipMemcpyAsync(vp(dst), vp(src), ctypes.c_size_t(n), kind, s))
def pipeline(d_a):
    def up(k):
        if k < nch:
            cp(d_a.value + k * C, h_in.ctypes.data + k * C, C, 1, s1)
            ck(hip.hipEventRecord(ev[k], s1))
    up(0); up(1)
    for k in range(nch):
        ck(hip.hipEventSynchronize(ev[k]))
        cp(h_out.ctypes.data + k * C, d_a.value + k * C, C, 2, s2)
        up(k + 2)
    ck(hip.hipStreamSynchronize(s1)); ck(hip.hipStreamSynchronize(s2))
def alloc():
    a = vp(); ck(hip.hipMalloc(ctypes.byref(a), ctypes.c_size_t(N))); return a
def reg(a): ck(hip.hipHostRegister(vp(a.ctypes.data), ctypes.c_size_t(N), 0))
def unreg(a): ck(hip.hipHostUnregister(vp(a.ctypes.data)))
if cfg != 'C': reg(h_in)
if cfg != 'D': reg(h_out)
d_a = alloc() if cfg != 'B' else None
if cfg == 'E':
    pipeline(d_a); unreg(h_in); unreg(h_out); reg(h_in); reg(h_out)
if cfg == 'F':
    pipeline(d_a); ck(hip.hipFree(d_a)); d_a = alloc()
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    if cfg == 'B': d_a = alloc()
    if cfg == 'C': reg(h_in)
    if cfg == 'D': reg(h_out)
    pipeline(d_a)
    if cfg == 'C': unreg(h_in)
    if cfg == 'D': unreg(h_out)
    if cfg == 'B': ck(hip.hipFree(d_a))
    ts.append((time.perf_counter() - t0) * 1e3)
print(cfg, ' '.join(f'{t:.2f}' for t in ts), 'ms', flush=True)
