"""The C-ABI library loads on a GPU-less machine and exports every symbol include/voltools_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from voltools_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'voltools_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vt_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/voltools_hip.h but not exported'
    assert sorted(_native.SYMBOLS) == names      # the ctypes shim binds exactly the declared set


def test_library_answers_without_a_gpu():
    lib = _native.load()
    assert lib.vt_version().decode().startswith('voltools_amd')
    n = _native.device_count()
    assert n >= 0
    if n == 0:
        # every compute entry point must fail loudly (non-zero code + message), never fall back to the CPU
        h = ctypes.c_void_p()
        data = np.zeros((4, 4, 4), np.float32)
        rc = lib.vt_volume_create(0, 4, 4, 4, 0, data.ctypes.data, 0, ctypes.byref(h))
        assert rc != 0 and lib.vt_last_error()
        with pytest.raises(RuntimeError):
            _native.check(rc, 'vt_volume_create')
        import voltools_amd as vt
        assert vt.AVAILABLE_DEVICES == ['cpu']
        with pytest.raises(ValueError):
            vt.affine(data, np.eye(4, dtype=np.float32), device='gpu')


def test_argument_validation_codes():
    lib = _native.load()
    h = ctypes.c_void_p()
    data = np.zeros((4, 4, 4), np.float32)
    assert lib.vt_volume_create(0, 0, 4, 4, 0, data.ctypes.data, 0, ctypes.byref(h)) != 0
    assert lib.vt_volume_create(0, 4, 4, 4, 9, data.ctypes.data, 0, ctypes.byref(h)) != 0
    assert lib.vt_volume_create(0, 4, 4, 4, 0, None, 0, ctypes.byref(h)) != 0
    assert lib.vt_volume_affine(None, data.ctypes.data, data.ctypes.data, 0) != 0
    assert lib.vt_volume_destroy(None) == 0
    n = ctypes.c_int(-1)
    assert lib.vt_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'voltools_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ('import oracle', 'from oracle', 'vt_oracle', 'libvt_oracle', 'oracle/'):
                    assert needle not in text, f'{f} references the oracle ({needle})'


def test_host_result_pool_reuses_only_released_buffers(monkeypatch):
    """_HostResultPool (results of calls without output=): a buffer is recycled only after the caller dropped every
    view of it; buffers the caller keeps are never handed out again; the cap evicts idle buffers."""
    import numpy as np
    from voltools_amd import _native

    class FakeLib:
        registered = set()

        def vt_host_register(self, dev, ptr, nbytes):
            self.registered.add(ptr.value)
            return 0

        def vt_host_unregister(self, dev, ptr):
            self.registered.discard(ptr.value)
            return 0

    fake = FakeLib()
    monkeypatch.setattr(_native, 'load', lambda: fake)
    monkeypatch.setenv('VT_HOST_POOL_MIN_MB', '1')
    pool = _native._HostResultPool()
    shape = (64, 64, 64)
    a = pool.take(shape, 0)
    b = pool.take(shape, 0)
    pa, pb = a.ctypes.data, b.ctypes.data
    assert pa != pb and a.shape == shape and a.dtype == np.float32 and {pa, pb} <= fake.registered
    del a
    c = pool.take(shape, 0)
    assert c.ctypes.data == pa                       # released -> recycled
    view = c[3:5]
    del c
    d = pool.take(shape, 0)
    assert d.ctypes.data not in (pa, pb)             # a live view keeps its buffer out of circulation
    small = pool.take((4, 4, 4), 0)
    assert small.ctypes.data not in fake.registered  # tiny results bypass the pool
    del view, b, d
    pool.cap = 3 * 64 ** 3 * 4
    e = pool.take((65, 64, 64), 0)
    assert sum(x[3] for x in pool.entries) <= pool.cap and e.shape == (65, 64, 64)
    pool.clear()
    assert not fake.registered
