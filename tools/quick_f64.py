#!/usr/bin/env python3
"""Device timing of the f64 path (Stockham kernel) on a cfg2-shaped batch:  python tools/quick_f64.py [n_clips] [nfft] [hop]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hop = int(sys.argv[3]) if len(sys.argv) > 3 else 256
N = 480000
_capi.ensure_device()
for dt, code, isz in ((np.float64, _capi.F64, 8), (np.float32, _capi.F32, 4)):
    x = (np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1).astype(dt)
    plan = _capi.Plan(n, n, hop, get_window(("tukey", 0.25), n), 1, 48000.0, 0, 0, code)
    if dt == np.float32:
        plan.force_kernel("stockham")
    nfr = plan.n_frames(N); nb = n // 2 + 1
    d_in = _capi.DeviceBuffer(x.nbytes); d_in.upload(x)
    d_out = _capi.DeviceBuffer(n_clips * nfr * nb * isz)
    fn = lambda: plan.stft(d_in.ptr, N, N, n_clips, d_out.ptr, nfr * nb)
    for _ in range(3): fn()
    _capi.stream_sync()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    _capi.stream_sync()
    dt_s = (time.perf_counter() - t0) / 10
    fr = n_clips * nfr
    print(f"{np.dtype(dt).name} kernel={plan.kernel} frames={fr} {dt_s*1e6:.0f} us  {fr/dt_s/1e6:.1f} M frames/s  {fr*(hop+nb)*isz/dt_s/1e9:.0f} GB/s algorithmic")
    d_in.free(); d_out.free(); plan.close()
