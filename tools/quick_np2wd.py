#!/usr/bin/env python3
"""Sustained timing of the wide double-precision register chirp-z kernel against the LDS chirp-z kernel of the same plan: python tools/quick_np2wd.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
_capi.ensure_device()
N, n_clips = 480000, 32
x = np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1
d_in = _capi.DeviceBuffer(x.nbytes); d_in.upload(x)
def leg(fn, secs=0.5):
    fn(); _capi.stream_sync()
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        fn(); k += 1
        if k % 8 == 0: _capi.stream_sync()
    _capi.stream_sync()
    return (time.perf_counter() - t0) / k
shapes = [(1056, 264), (2000, 500), (2000, 1750), (3000, 750), (4000, 1000), (6000, 1500), (8000, 2000), (8160, 7140)]
for n, hop in shapes:
    plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F64)
    nf = plan.n_frames(N)
    out = _capi.DeviceBuffer(n_clips * nf * (n // 2 + 1) * 8)
    bp = _capi.DeviceBuffer(n_clips * nf * 8)
    fn = lambda: plan.stft(d_in.ptr, N, N, n_clips, out.ptr, nf * (n // 2 + 1))
    fb = lambda: plan.band_power(d_in.ptr, N, N, n_clips, 1, 40, bp.ptr, nf)
    res = {}
    for k in (plan.kernel, "bluestein"):
        plan.force_kernel(k)
        res[k] = leg(fn)
    plan.force_kernel("rbluewd")
    tb = leg(fb)
    fr = n_clips * nf
    tw, tl = res["rbluewd"], res["bluestein"]
    byts = fr * (hop * 8 + (n // 2 + 1) * 8)
    print(f"n{n} hop {hop}: rbluewd {tw*1e6:9.1f} us ({fr/tw/1e6:7.1f} M frames/s, {byts/tw/1e12:.2f} TB/s algorithmic)  band {tb*1e6:9.1f} us   | LDS chirp-z {tl*1e6:9.1f} us ({fr/tl/1e6:6.1f} M frames/s)  -> x{tl/tw:.1f}", flush=True)
    out.free(); bp.free(); plan.close()
