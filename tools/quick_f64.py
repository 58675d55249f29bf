#!/usr/bin/env python3
"""Device timing of the f64 path on a cfg2-shaped batch, default kernel vs the LDS Stockham kernel (sustained: >= QB_SECS per leg):
python tools/quick_f64.py [n_clips] [nfft] [hop]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hop = int(sys.argv[3]) if len(sys.argv) > 3 else 256
secs = float(os.environ.get("QB_SECS", "1.0"))
N = 480000
_capi.ensure_device()
legs = ((np.float64, _capi.F64, 8, None), (np.float64, _capi.F64, 8, "stockham"), (np.float32, _capi.F32, 4, None))
if os.environ.get("QF_LEGS"):                                # e.g. QF_LEGS=0 under tools/telemetry.py
    legs = [legs[int(i)] for i in os.environ["QF_LEGS"].split(",")]
for dt, code, isz, force in legs:
    x = (np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1).astype(dt)
    plan = _capi.Plan(n, n, hop, get_window(("tukey", 0.25), n), 1, 48000.0, 0, 0, code)
    if force:
        plan.force_kernel(force)
    nfr = plan.n_frames(N); nb = n // 2 + 1
    d_in = _capi.DeviceBuffer(x.nbytes); d_in.upload(x)
    d_out = _capi.DeviceBuffer(n_clips * nfr * nb * isz)
    fn = lambda: plan.stft(d_in.ptr, N, N, n_clips, d_out.ptr, nfr * nb)
    for _ in range(3): fn()
    _capi.stream_sync()
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(20): fn()
        _capi.stream_sync()
        reps += 20
    dt_s = (time.perf_counter() - t0) / reps
    fr = n_clips * nfr
    print(f"{np.dtype(dt).name} kernel={plan.kernel} frames={fr} {dt_s*1e6:.0f} us  {fr/dt_s/1e6:.1f} M frames/s  {fr*(hop+nb)*isz/dt_s/1e9:.0f} GB/s algorithmic")
    d_in.free(); d_out.free(); plan.close()
