#!/usr/bin/env python3
"""Time the A8-A13 epilogue kernels on the cfg2-size spectrum (119 808 x 513 f32 = 246 MB) against their HBM bytes.
Three spectra rotate so that the 256 MiB Infinity Cache cannot serve the reads."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
_capi.ensure_device()
L = _capi.lib()
rows, nb = 119808, 513
host = (np.random.default_rng(0).random((rows, nb), dtype=np.float32) + 1e-3)
specs = [_capi.DeviceBuffer(rows * nb * 4) for _ in range(3)]
for b_ in specs:
    b_.upload(host)
_capi.stream_sync()
turn = [0]


class _Rot:
    @property
    def ptr(self):
        turn[0] += 1
        return specs[turn[0] % 3].ptr


spec = _Rot()
img = _capi.DeviceBuffer(rows * nb * 4)
band = _capi.DeviceBuffer(rows * 4)
feat = _capi.DeviceBuffer(rows * 8)
mm = _capi.DeviceBuffer(64)
sums = _capi.DeviceBuffer(8 * 16)
def timed(fn, iters=10):
    fn(); _capi.stream_sync()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    _capi.stream_sync()
    return (time.perf_counter() - t0) / iters
P = lambda b: C.c_void_p(b.ptr)
full = rows * nb * 4
lo7 = (C.c_int * 7)(0, 0, 10, 20, 40, 100, 300); hi7 = (C.c_int * 7)(513, 10, 20, 40, 100, 300, 513)
cases = {
 "minmax full": (lambda: L.sg_minmax(P(spec), 0, rows, nb, 0, 512, P(mm), None), full),
 "normalise dB full": (lambda: L.sg_normalise_image(P(spec), 0, rows, nb, 0, 512, 1, 0.0, P(img), P(mm), None), 3 * full),
 "normalise lin band 0..63": (lambda: L.sg_normalise_image(P(spec), 0, rows, nb, 0, 63, 0, 0.0, P(img), P(mm), None), 3 * rows * 64 * 4),
 "slice band 0..63": (lambda: L.sg_slice_bins(P(spec), 0, rows, nb, 0, 63, P(img), None), 2 * rows * 64 * 4),
 "band_sum 5..200": (lambda: L.sg_band_sum(P(spec), 0, rows, nb, 5, 200, P(band), None), rows * 196 * 4),
 "band_features": (lambda: L.sg_band_features(P(band), 0, rows, P(feat), None), rows * 12),
 "band_totals 7 ranges": (lambda: L.sg_band_totals(P(spec), 0, rows, nb, 7, lo7, hi7, P(sums), None), full),
}
for name, (fn, nbytes) in cases.items():
    t = timed(fn)
    print(f"{name:28s} {t*1e6:8.1f} us   {nbytes/t/1e9:7.0f} GB/s (algorithmic)")
