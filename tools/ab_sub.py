#!/usr/bin/env python3
"""Round 3: the headline launch with its frames walked as `sub` interleaved sequences (SPECTRO_R8_SUB, read per launch): at hop 256, sub = 2 makes two waves
write alternate rows of one region (each sliding its window by 512 samples), sub = 4 four waves (no sliding left).  Same box, interleaved legs, four rotating
buffer sets, output compared bit for bit with sub = 1.      python tools/ab_sub.py [hop] [secs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
n_clips, N, n = 64, 480000, 1024
_capi.ensure_device()
x = (np.random.default_rng(1234).standard_normal((n_clips, N)) * 0.1).astype(np.float32)
plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F32)
nfr, nb = plan.n_frames(N), n // 2 + 1
NB = 4
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(NB)]
outs = [_capi.DeviceBuffer(n_clips * nfr * nb * 4) for _ in range(NB)]
for b in ins: b.upload(x)
_capi.stream_sync()
subs = [s for s in (1, 2, 4, 8) if s == 1 or hop * s in (128, 256, 512, 896)]      # the register-sliding instances (launch_one)
ref = None
for s in subs:
    os.environ["SPECTRO_R8_SUB"] = str(s)
    plan.stft(ins[0].ptr, N, N, n_clips, outs[0].ptr, nfr * nb); _capi.stream_sync()
    got = np.empty((n_clips, nfr, nb), np.float32); outs[0].download(got); _capi.stream_sync()
    if ref is None: ref = got
    print(f"sub {s}: output {'identical to sub 1' if np.array_equal(ref, got) else 'DIFFERS from sub 1'}", flush=True)
for rep in range(3):
    for s in subs:
        os.environ["SPECTRO_R8_SUB"] = str(s)
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < secs:
            for i in range(20): plan.stft(ins[(reps + i) % NB].ptr, N, N, n_clips, outs[(reps + i) % NB].ptr, nfr * nb)
            _capi.stream_sync(); reps += 20
        dt = (time.perf_counter() - t0) / reps
        print(f"hop {hop} sub {s}: {dt*1e6:.1f} us per launch  {n_clips*nfr/dt/1e9:.3f} G frames/s", flush=True)
