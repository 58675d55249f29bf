#!/usr/bin/env python3
"""Where the host-to-host time of spectro.spectrogram goes on the cfg2 batch (for DESIGN.md, PCIe-inclusive path)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
_capi.ensure_device()
x = (np.random.default_rng(1234).standard_normal((64, 480000)) * 0.1).astype(np.float32)
plan = _capi.Plan(1024, 1024, 256, get_window("hann", 1024), 1, 48000.0, 0, 0, _capi.F32)
nfr = plan.n_frames(480000)
def T(label, fn, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); _capi.stream_sync(); best = min(best, time.perf_counter() - t0)
    print(f"{label:50s} {best*1e3:8.2f} ms"); return r
out_bytes = 64 * nfr * 513 * 4
bufs = T("sg_malloc in + out", lambda: (_capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(out_bytes)))
d_in, d_out = bufs
T("H2D 123 MB from a numpy array (pageable)", lambda: d_in.upload(x))
T("kernel", lambda: plan.stft(d_in.ptr, 480000, 480000, 64, d_out.ptr, nfr * 513))
T("np.empty(246 MB) + D2H (fresh pages)", lambda: d_out.download(np.empty((64, nfr, 513), np.float32)))
warm = np.zeros((64, nfr, 513), np.float32)
T("D2H into an already-touched array", lambda: d_out.download(warm))
T("np.zeros(246 MB) alone (page faults only)", lambda: np.zeros((64, nfr, 513), np.float32) + 0)
T("sg_free both", lambda: (d_in.free(), d_out.free()), n=1)
