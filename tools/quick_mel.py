#!/usr/bin/env python3
"""Quick device timing of the standalone mel epilogue (cfg3 shape):  [SPECTRO_LIB=...] python tools/quick_mel.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.mel import MelBank
_capi.ensure_device()
frames = 119808
spec = _capi.DeviceBuffer(frames * 513 * 4)
spec.upload(np.random.default_rng(0).random((frames, 513), dtype=np.float32)); _capi.stream_sync()
d_mel = _capi.DeviceBuffer(frames * 80 * 4)
bank = MelBank(1024, 48000.0, 80, 0.0, 24000.0)
for dense in (False, True):
    fn = lambda: bank.apply_ptr(spec.ptr, frames, d_mel.ptr, True, dense=dense)
    for _ in range(5): fn()
    _capi.stream_sync()
    t0 = time.perf_counter()
    for _ in range(50): fn()
    _capi.stream_sync()
    dt = (time.perf_counter() - t0) / 50
    print(f"mel {'dense' if dense else 'block-sparse'} {dt*1e6:.1f} us  {frames*(513+80)*4/dt/1e9:.0f} GB/s")
