#!/usr/bin/env python3
"""Sustained timing of the LDS chirp-z kernel (the sizes no register kernel takes): python tools/quick_bluestein.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
_capi.ensure_device()
N, n_clips = 480000, 16
for dt, code, isz in ((np.float32, _capi.F32, 4), (np.float64, _capi.F64, 8)):
    x = (np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1).astype(dt)
    d_in = _capi.DeviceBuffer(x.nbytes); d_in.upload(x)
    for n, hop in ((3000, 750), (6000, 1500), (2080, 520), (1001, 250), (1500, 375)):
        plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, code)
        if plan.kernel != "bluestein":
            plan.close(); continue
        nf = plan.n_frames(N)
        out = _capi.DeviceBuffer(n_clips * nf * (n // 2 + 1) * isz)
        fn = lambda: plan.stft(d_in.ptr, N, N, n_clips, out.ptr, nf * (n // 2 + 1))
        fn(); _capi.stream_sync()
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            fn(); k += 1; _capi.stream_sync()
        dtm = (time.perf_counter() - t0) / k
        print(f"{np.dtype(dt).name} n{n} hop {hop}: {dtm*1e6:9.1f} us per {n_clips * nf} frames = {n_clips*nf/dtm/1e6:6.1f} M frames/s", flush=True)
        out.free(); plan.close()
    d_in.free()
