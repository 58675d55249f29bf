#!/usr/bin/env python3
"""Pieces of the host<->device path on the cfg2 batch: pageable / pinned / registered copies, one and two streams."""
import ctypes as C, gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.pipeline import pinned_empty
_capi.ensure_device()
L = _capi.lib()
x = (np.random.default_rng(1).standard_normal((64, 480000)) * 0.1).astype(np.float32)      # 123 MB
out_shape = (64, 1872, 513)
nb_out = int(np.prod(out_shape)) * 4
d_in, d_out = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(nb_out)


def t(name, fn, reps=4):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); _capi.stream_sync(); best = min(best, time.perf_counter() - t0)
    print(f"{name:64s} {best*1e3:8.2f} ms")
    return best


t0 = time.perf_counter(); po = pinned_empty(out_shape, np.float32); print(f"pinned_empty 246 MB (first, hipHostMalloc)                        {(time.perf_counter()-t0)*1e3:8.2f} ms")
del po; gc.collect()
t0 = time.perf_counter(); po = pinned_empty(out_shape, np.float32); print(f"pinned_empty 246 MB (pooled)                                       {(time.perf_counter()-t0)*1e3:8.2f} ms")
px = pinned_empty(x.shape, np.float32); px[:] = x
t("H2D 123 MB from pageable numpy", lambda: d_in.upload(x))
t("H2D 123 MB from pinned", lambda: d_in.upload(px))
pg = np.empty(out_shape, np.float32)
t("D2H 246 MB into fresh np.empty (first touch each time)", lambda: d_out.download(np.empty(out_shape, np.float32)), reps=3)
t("D2H 246 MB into touched pageable", lambda: d_out.download(pg))
t("D2H 246 MB into pinned", lambda: d_out.download(po))
# two streams, pinned both ways, 8 chunks
s1, s2 = C.c_void_p(), C.c_void_p()
_capi.check(L.sg_stream_create(C.byref(s1))); _capi.check(L.sg_stream_create(C.byref(s2)))


def duplex(src):
    n = 8
    ci, co = x.nbytes // n, nb_out // n
    for i in range(n):
        s = s1 if i % 2 == 0 else s2
        _capi.check(L.sg_memcpy_h2d(C.c_void_p(d_in.ptr + i * ci), C.c_void_p(src.ctypes.data + i * ci), ci, s))
        _capi.check(L.sg_memcpy_d2h(C.c_void_p(po.ctypes.data + i * co), C.c_void_p(d_out.ptr + i * co), co, s))
    _capi.check(L.sg_stream_sync(s1)); _capi.check(L.sg_stream_sync(s2))


t("8 chunks on 2 streams: H2D pinned + D2H pinned (duplex)", lambda: duplex(px))
t("8 chunks on 2 streams: H2D pageable + D2H pinned", lambda: duplex(x))
t0 = time.perf_counter(); _capi.check(L.sg_host_register(C.c_void_p(x.ctypes.data), x.nbytes)); print(f"sg_host_register 123 MB                                            {(time.perf_counter()-t0)*1e3:8.2f} ms")
t("H2D 123 MB from registered numpy", lambda: d_in.upload(x))
t("8 chunks on 2 streams: H2D registered + D2H pinned", lambda: duplex(x))
t0 = time.perf_counter(); _capi.check(L.sg_host_unregister(C.c_void_p(x.ctypes.data))); print(f"sg_host_unregister                                                 {(time.perf_counter()-t0)*1e3:8.2f} ms")
