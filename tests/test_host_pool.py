"""The pooled host result buffers (voltools_amd/_native.py::_HostResultPool) without a GPU: page alignment of everything that
gets registered, reuse only after the caller has dropped every view, eviction.  The library is replaced by a recorder."""
import ctypes

import numpy as np

from voltools_amd import _native


class _Recorder:
    def __init__(self):
        self.live = {}

    def vt_host_register(self, dev, ptr, nbytes):
        assert ptr.value % (2 << 20) == 0 and nbytes % (2 << 20) == 0, 'registered ranges must be whole 2 MiB units (transparent huge pages)'
        for a, n in self.live.items():
            assert ptr.value + nbytes <= a or a + n <= ptr.value, 'registered ranges must not share pages'
        self.live[ptr.value] = nbytes
        return 0

    def vt_host_unregister(self, dev, ptr):
        del self.live[ptr.value]
        return 0


def test_pool_buffers_are_page_isolated_and_reused_only_when_released(monkeypatch):
    rec = _Recorder()
    monkeypatch.setattr(_native, '_lib', rec)
    monkeypatch.setenv('VT_HOST_POOL_MB', '8')
    monkeypatch.setenv('VT_HOST_POOL_MIN_MB', '1')
    pool = _native._HostResultPool()
    shape = (64, 64, 80)                                   # 1.25 MiB
    a = pool.take(shape, 0)
    assert a.shape == shape and a.dtype == np.float32 and a.ctypes.data % 4096 == 0 and a.flags.c_contiguous
    b = pool.take(shape, 0)
    assert b.ctypes.data != a.ctypes.data                  # `a` is still held by the caller
    pa = a.ctypes.data
    keep = a[3]                                            # a slice keeps the buffer alive as well
    del a
    c = pool.take(shape, 0)
    assert c.ctypes.data not in (pa, b.ctypes.data)
    del keep
    d = pool.take(shape, 0)
    assert d.ctypes.data == pa                             # released -> reused
    # other sizes push the total over the 8 MiB cap: free buffers are unregistered, held ones never
    del d
    big = [pool.take((64, 64, 200), 0) for _ in range(3)]  # 3 x 3.1 MiB
    assert all(x.ctypes.data % 4096 == 0 for x in big)
    assert sum(rec.live.values()) <= (8 << 20) + (4 << 20)                  # (registered bytes are 2 MiB-granular: 2 x 2 + 4 MiB here)
    assert b.ctypes.data in rec.live and c.ctypes.data in rec.live
    small = pool.take((8, 8, 8), 0)                        # below the pool threshold: a plain array
    assert small.shape == (8, 8, 8)
    pool.clear()
    assert not rec.live
