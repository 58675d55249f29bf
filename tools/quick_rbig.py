#!/usr/bin/env python3
"""Sustained timing of the rbig kernels (nfft 2048 / 4096) at the sweep's hops, spectrum and fused band power:
python tools/quick_rbig.py     (SPECTRO_RBIG_NO_SLIDE=1 | 4 for the A/B)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
_capi.ensure_device()
N, n_clips = 480000, 64
x = (np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1).astype(np.float32)
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(4)]
for b in ins: b.upload(x)
secs = float(os.environ.get("QB_SECS", "0.6"))
for n in [int(v) for v in os.environ.get("QR_SIZES", "2048,4096").split(",")]:
    for hop in (64, 128, 256):
        plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F32)
        nf = plan.n_frames(N)
        outs = [_capi.DeviceBuffer(n_clips * nf * (n // 2 + 1) * 4) for _ in range(2)]
        d_bp = _capi.DeviceBuffer(n_clips * nf * 4)
        res = []
        for name, fn in (("spectrum", lambda i: plan.stft(ins[i % 4].ptr, N, N, n_clips, outs[i % 2].ptr, nf * (n // 2 + 1))),
                         ("band", lambda i: plan.band_power(ins[i % 4].ptr, N, N, n_clips, 1, n // 4, d_bp.ptr, nf))):
            for i in range(3): fn(i)
            _capi.stream_sync()
            k, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < secs:
                for _ in range(5): fn(k); k += 1
                _capi.stream_sync()
            res.append((time.perf_counter() - t0) / k)
        print(f"n{n} hop {hop}: spectrum {res[0]*1e3:.3f} ms ({n_clips*nf/res[0]/1e9:.3f} G frames/s)  band {res[1]*1e3:.3f} ms", flush=True)
        for o in outs: o.free()
        d_bp.free(); plan.close()
