#!/usr/bin/env python3
"""Host-to-host latency of the drop-in call on GUI-sized inputs (BASELINE cfg1 and a 60 s EEG sweep) next to scipy."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
import spectro
from scipy.signal import spectrogram as sp_spectrogram

def bench(fn, n=200):
    fn(); fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[int(len(ts) * 0.99)] * 1e6

rng = np.random.default_rng(0)
cases = {
    "cfg1 f64 N=16000 nperseg=512": (rng.standard_normal(16000), 16000.0, 512),
    "cfg1 f32 N=16000 nperseg=512": (rng.standard_normal(16000).astype(np.float32), 16000.0, 512),
    "EEG f64 N=30000 nperseg=256 (GUI default)": (rng.standard_normal(30000), 500.0, 256),
    "1 clip f32 N=480000 nperseg=1024": (rng.standard_normal(480000).astype(np.float32), 48000.0, 1024),
}
for name, (x, fs, n) in cases.items():
    g = bench(lambda: spectro.spectrogram(x, fs=fs, nperseg=n, scaling="density", mode="psd"))
    c = bench(lambda: sp_spectrogram(x, fs=fs, nperseg=n, scaling="density", mode="psd"), 50)
    print(f"{name:45s} device path {g[0]:8.1f} us (p99 {g[1]:8.1f})   scipy {c[0]:8.1f} us (p99 {c[1]:8.1f})")

# the reference's second call site (PlotEngine.py:232): band features of one sweep
from spectro import engine
import scipy.signal as ss
for name, (x, fs, n) in cases.items():
    g = bench(lambda: engine.band_features(x, fs, n, 5.0, 30.0 if fs < 1000 else 3000.0))
    def ref():
        f, t, s = ss.spectrogram(x, fs=fs, nperseg=n, scaling="density", mode="psd")
        m = (f >= 5.0) & (f <= (30.0 if fs < 1000 else 3000.0))
        lp = np.log10(s[m].sum(axis=0) + 1e-20)
        return np.column_stack([lp, np.diff(lp, prepend=lp[0])])
    c = bench(ref, 50)
    print(f"features: {name:35s} device path {g[0]:8.1f} us (p99 {g[1]:8.1f})   scipy+numpy {c[0]:8.1f} us (p99 {c[1]:8.1f})")
