#!/usr/bin/env python3
"""Sustained timing of non-power-of-two nperseg (register chirp-z kernel against the LDS one): python tools/quick_np2.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
_capi.ensure_device()
N, n_clips = 480000, 64
f64 = os.environ.get("QN_DTYPE", "f32") == "f64"          # QN_DTYPE=f64: the double-precision kernel (nperseg <= 1024)
dt, code, isz = (np.float64, _capi.F64, 8) if f64 else (np.float32, _capi.F32, 4)
x = (np.random.default_rng(1).standard_normal((n_clips, N)) * 0.1).astype(dt)
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(4)]
for b in ins: b.upload(x)
secs = float(os.environ.get("QB_SECS", "0.5"))
shapes = [(1000, 250), (960, 240), (96, 24), (480, 120), (1504, 376), (2016, 504), (1000, 876)]
if f64:
    shapes = [(1000, 250), (960, 240), (96, 24), (480, 120), (1000, 875)]
for n, hop in shapes:
    plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, code)
    nf = plan.n_frames(N)
    outs = [_capi.DeviceBuffer(n_clips * nf * (n // 2 + 1) * isz) for _ in range(2)]
    d_bp = _capi.DeviceBuffer(n_clips * nf * isz)
    res = {}
    for kern in (plan.kernel, "bluestein"):
        plan.force_kernel(kern)
        fns = [("spectrum", lambda i: plan.stft(ins[i % 4].ptr, N, N, n_clips, outs[i % 2].ptr, nf * (n // 2 + 1)))]
        if kern != "bluestein":
            fns.append(("band", lambda i: plan.band_power(ins[i % 4].ptr, N, N, n_clips, 1, n // 4, d_bp.ptr, nf)))
        for name, fn in fns:
            for i in range(2): fn(i)
            _capi.stream_sync()
            k, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < (secs if kern != "bluestein" else 0.25):
                for _ in range(2 if kern == "bluestein" else 5): fn(k); k += 1
                _capi.stream_sync()
            res[(kern, name)] = (time.perf_counter() - t0) / k
    fr = n_clips * nf
    k0 = [k for k in res if k[0] != "bluestein" and k[1] == "spectrum"][0]
    bpf = hop * isz + (n // 2 + 1) * isz
    print(f"n{n} hop {hop}: {k0[0]} {res[k0]*1e6:8.1f} us ({fr/res[k0]/1e6:7.1f} M frames/s, {fr*bpf/res[k0]/1e12:.2f} TB/s algorithmic)  band {res.get((k0[0], 'band'), 0)*1e6:8.1f} us"
          f"   | LDS chirp-z {res[('bluestein', 'spectrum')]*1e6:9.1f} us ({fr/res[('bluestein', 'spectrum')]/1e6:6.1f} M frames/s)  -> x{res[('bluestein', 'spectrum')]/res[k0]:.1f}", flush=True)
    for o in outs: o.free()
    d_bp.free(); plan.close()
