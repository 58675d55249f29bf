#!/usr/bin/env python3
"""Quick device timing of sg_stft on the cfg2 shape (no torch): python tools/quick_bench.py [n_clips] [hop] [kernel|-] [nfft] [n_samples]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi  # noqa: E402
from spectro.windows import get_window  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kernel = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
N, n = (int(sys.argv[5]) if len(sys.argv) > 5 else 480000), (int(sys.argv[4]) if len(sys.argv) > 4 else 1024)
_capi.ensure_device()
print(_capi.device_info())
x = (np.random.default_rng(1234).standard_normal((n_clips, N)) * 0.1).astype(np.float32)
plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F32)
if kernel:
    plan.force_kernel(kernel)
nfr = plan.n_frames(N)
NB = 4   # rotate buffer sets so the 256 MiB infinity cache cannot hold the working set
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(NB)]
outs = [_capi.DeviceBuffer(n_clips * nfr * (n // 2 + 1) * 4) for _ in range(NB)]
for b in ins:
    b.upload(x)
_capi.stream_sync()
for i in range(3):
    plan.stft(ins[i % NB].ptr, N, N, n_clips, outs[i % NB].ptr, nfr * (n // 2 + 1))
_capi.stream_sync()
frames = n_clips * nfr
secs = float(os.environ.get("QB_SECS", "0"))       # QB_SECS=1.5: sustained run (clocks and board power settle) before the timed repetitions
if secs > 0:
    t_s, i = time.perf_counter(), 0
    while time.perf_counter() - t_s < secs:
        for _ in range(16):
            plan.stft(ins[i % NB].ptr, N, N, n_clips, outs[i % NB].ptr, nfr * (n // 2 + 1))
            i += 1
        _capi.stream_sync()
for rep in range(3):
    t0 = time.perf_counter()
    iters = 40
    for i in range(iters):
        plan.stft(ins[i % NB].ptr, N, N, n_clips, outs[i % NB].ptr, nfr * (n // 2 + 1))
    _capi.stream_sync()
    dt = (time.perf_counter() - t0) / iters
    bpf = hop * 4 + (n // 2 + 1) * 4
    print(f"kernel={plan.kernel} clips={n_clips} hop={hop} frames={frames} {dt*1e6:.1f} us/launch "
          f"{frames/dt/1e6:.1f} Mframes/s  {frames*bpf/dt/1e9:.0f} GB/s algorithmic ({frames*bpf/dt/8e12*100:.1f}% of 8 TB/s)")
ms = plan.time_stft(ins[0].ptr, N, N, n_clips, outs[0].ptr, nfr * (n // 2 + 1), 20)
print(f"hipEvent single-buffer: {ms*1e3:.1f} us/launch")
