#!/usr/bin/env python3
"""Sustained timing of whatever kernel a plan picks: python tools/quick_any.py 8192:2048:f32 64:16:f64 ... (nperseg:hop:dtype[:clips]; QA_SECS = seconds per leg, default 0.5)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
_capi.ensure_device()
N = 480000
bufs = {}
for spec in sys.argv[1:]:
    parts = spec.split(":")
    n, hop, dt = int(parts[0]), int(parts[1]), parts[2]
    n_clips = int(parts[3]) if len(parts) > 3 else 32
    code, npdt, isz = (_capi.F32, np.float32, 4) if dt == "f32" else (_capi.F64, np.float64, 8)
    key = (dt, n_clips)
    if key not in bufs:
        x = (np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1).astype(npdt)
        b = _capi.DeviceBuffer(x.nbytes); b.upload(x); bufs[key] = b
    d_in = bufs[key]
    plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, code)
    nf = plan.n_frames(N)
    out = _capi.DeviceBuffer(n_clips * nf * (n // 2 + 1) * isz)
    fn = lambda: plan.stft(d_in.ptr, N, N, n_clips, out.ptr, nf * (n // 2 + 1))
    fn(); _capi.stream_sync()
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ.get("QA_SECS", "0.5")):
        fn(); k += 1
        if k % 8 == 0: _capi.stream_sync()
    _capi.stream_sync()
    t = (time.perf_counter() - t0) / k
    fr = n_clips * nf
    print(f"{dt} n{n} hop {hop} [{plan.kernel}]: {t*1e6:9.1f} us per {fr} frames = {fr/t/1e6:8.1f} M frames/s, {fr*(hop+n//2+1)*isz/t/1e12:.2f} TB/s algorithmic", flush=True)
    out.free(); plan.close()
