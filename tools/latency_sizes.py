#!/usr/bin/env python3
"""GUI-sized calls (one 10 s sweep at 20 kHz, float64, the reference's literal call) across the spin box's range: first call (plan + tables) and steady state, host numpy -> numpy.
python tools/latency_sizes.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
import spectro
from spectro import _capi
from spectro.signal import plan_for
_capi.ensure_device()
x = np.random.default_rng(0).standard_normal(200000) * 0.1
spectro.spectrogram(x, fs=20000.0, nperseg=1024)
for dt in (np.float64, np.float32):
    xs = x.astype(dt)
    for n in (32, 64, 128, 256, 512, 992, 1024, 2048, 3008, 4096, 6016, 8160, 8192):
        t0 = time.perf_counter()
        f, t, s = spectro.spectrogram(xs, fs=20000.0, nperseg=n, scaling="density", mode="psd")
        first = time.perf_counter() - t0
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            spectro.spectrogram(xs, fs=20000.0, nperseg=n, scaling="density", mode="psd")
            ts.append(time.perf_counter() - t0)
        print(f"{np.dtype(dt).name} nperseg {n:5d}: {s.shape[1]:5d} frames, first call {first*1e3:7.2f} ms, then p50 {np.median(ts)*1e3:6.3f} ms  p90 {np.quantile(ts, 0.9)*1e3:6.3f} ms", flush=True)
