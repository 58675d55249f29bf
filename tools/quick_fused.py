#!/usr/bin/env python3
"""Sustained device timing of the fused STFT+mel call (cfg3 shape):  [SPECTRO_FUSED_V1=1] python tools/quick_fused.py [secs]
Runs back-to-back launches for `secs` (default 1.5) after 0.4 s of warm-up, so the clocks are where a batch job holds them
(a 25 ms burst is timed at the idle clock: 0.3-1 GHz), rotating four input buffers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi
from spectro.mel import MelBank
from spectro.windows import get_window
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
n_mels = int(sys.argv[2]) if len(sys.argv) > 2 else 80
_capi.ensure_device()
N, n_clips = 480000, 64
x = (np.random.default_rng(1234).standard_normal((n_clips, N)) * 0.1).astype(np.float32)
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(4)]
for b in ins:
    b.upload(x)
plan = _capi.Plan(1024, 1024, 256, get_window("hann", 1024), 1, 48000.0, 0, 0, _capi.F32)
nfr = plan.n_frames(N)
d_mel = _capi.DeviceBuffer(n_clips * nfr * n_mels * 4)
bank = MelBank(1024, 48000.0, n_mels, 0.0, 24000.0)
fn = lambda i: bank.stft_mel_ptr(plan, ins[i % 4].ptr, N, N, n_clips, d_mel.ptr, nfr * n_mels, True)
_capi.stream_sync()


def run(duration):
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < duration:
        for _ in range(32):
            fn(n)
            n += 1
        _capi.stream_sync()
    return (time.perf_counter() - t0) / n


run(0.4)
dt = run(secs)
print(f"fused [{'mfma' if (bank._sparse is None or os.environ.get('SPECTRO_FUSED_MFMA') == '1') else 'sparse ipl=%d' % bank._sparse[0]}] {dt*1e6:.1f} us  {n_clips*nfr/dt/1e9:.3f} G frames/s  ({n_mels} mels, sustained {secs:g} s)")
