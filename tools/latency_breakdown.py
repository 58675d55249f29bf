#!/usr/bin/env python3
"""Where a GUI-sized spectro.spectrogram call spends its time (cfg1 f64: N = 16000, nperseg 512)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
import spectro
from spectro import _capi, signal as sig
x = np.random.default_rng(0).standard_normal(16000)
kw = dict(fs=16000.0, nperseg=512, scaling="density", mode="psd")
for _ in range(20): spectro.spectrogram(x, **kw)
def med(fn, n=300):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts)//2] * 1e6
print(f"whole call                 {med(lambda: spectro.spectrogram(x, **kw)):7.1f} us")
win, nperseg = sig.resolve_segments(("tukey", .25), 512, 16000)
print(f"resolve_segments (window)  {med(lambda: sig.resolve_segments(('tukey', .25), 512, 16000)):7.1f} us")
plan = sig.plan_for(win, 512, 512, 448, 1, 16000.0, 0, 0, _capi.F64)
print(f"plan_for (cache hit)       {med(lambda: sig.plan_for(win, 512, 512, 448, 1, 16000.0, 0, 0, _capi.F64)):7.1f} us")
nfr = plan.n_frames(16000)
out = np.empty((1, nfr, 257))
print(f"2 x DeviceBuffer + free    {med(lambda: (_capi.DeviceBuffer(x.nbytes).free(), _capi.DeviceBuffer(out.nbytes).free())):7.1f} us")
d_in, d_out = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(out.nbytes)
print(f"upload + sync              {med(lambda: (d_in.upload(x), _capi.stream_sync())):7.1f} us")
print(f"kernel + sync              {med(lambda: (plan.stft(d_in.ptr, 16000, 16000, 1, d_out.ptr, nfr * 257), _capi.stream_sync())):7.1f} us")
print(f"download + sync            {med(lambda: (d_out.download(out), _capi.stream_sync())):7.1f} us")
print(f"upload+kernel+download+sync{med(lambda: (d_in.upload(x), plan.stft(d_in.ptr, 16000, 16000, 1, d_out.ptr, nfr * 257), d_out.download(out), _capi.stream_sync())):7.1f} us")
print(f"freqs + times              {med(lambda: (_capi.freqs(512, 16000.0), _capi.times(16000, 512, 448, 16000.0))):7.1f} us")
print(f"np.empty + ascontiguous    {med(lambda: (np.empty((1, nfr, 257)), np.ascontiguousarray(x.reshape(1, 16000)))):7.1f} us")
