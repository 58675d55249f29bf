#!/usr/bin/env python3
"""Device timing of the fused band-power path (A11, sg_stft_band_power) on the cfg2 batch:  python tools/quick_band.py [hop]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_capi.ensure_device()
N, n_clips = 480000, 64
x = (np.random.default_rng(1234).standard_normal((n_clips, N)) * 0.1).astype(np.float32)
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(4)]
for b in ins: b.upload(x)
plan = _capi.Plan(1024, 1024, hop, get_window("hann", 1024), 1, 48000.0, 0, 0, _capi.F32)
nfr = plan.n_frames(N)
band = _capi.DeviceBuffer(n_clips * nfr * 4)
turn = [0]
def fn():
    turn[0] += 1
    plan.band_power(ins[turn[0] % 4].ptr, N, N, n_clips, 10, 200, band.ptr, nfr)
for _ in range(20): fn()
_capi.stream_sync()
t0 = time.perf_counter()
for _ in range(200): fn()
_capi.stream_sync()
dt = (time.perf_counter() - t0) / 200
fr = n_clips * nfr
print(f"band power hop={hop}: {dt*1e6:.1f} us  {fr/dt/1e9:.3f} G frames/s  input {fr*hop*4/dt/1e9:.0f} GB/s")
