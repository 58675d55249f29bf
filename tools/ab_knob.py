#!/usr/bin/env python3
"""One launch-time knob of the headline kernel (an environment variable the library reads per launch: SPECTRO_R8_WAVES, SPECTRO_R8_OCC,
SPECTRO_R8_SUB ...) A/B'd inside ONE process: legs interleaved, four rotating buffer sets, outputs compared bit for bit with the first value
('-' = variable unset).      python tools/ab_knob.py VAR hop secs value [value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.windows import get_window
var, hop, secs, values = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4:]
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

def setv(v):
    if v == "-": os.environ.pop(var, None)
    else: os.environ[var] = v

ref = None
for v in values:
    setv(v)
    plan.stft(ins[0].ptr, N, N, n_clips, outs[0].ptr, nfr * nb); _capi.stream_sync()
    got = np.empty((n_clips, nfr, nb), np.float32); outs[0].download(got); _capi.stream_sync()
    if ref is None: ref = got
    print(f"{var}={v}: output {'identical to the first leg' if np.array_equal(ref, got) else 'DIFFERS from the first leg'}", flush=True)
for rep in range(3):
    for v in values:
        setv(v)
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < secs:
            for i in range(20): plan.stft(ins[(reps + i) % NB].ptr, N, N, n_clips, outs[(reps + i) % NB].ptr, nfr * nb)
            _capi.stream_sync(); reps += 20
        dt = (time.perf_counter() - t0) / reps
        print(f"hop {hop} {var}={v}: {dt*1e6:.1f} us per launch  {n_clips*nfr/dt/1e9:.3f} G frames/s", flush=True)
