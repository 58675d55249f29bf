#!/usr/bin/env python3
"""End-to-end numpy -> numpy time of the cfg2 batch (PCIe-inclusive), unpipelined and pipelined, for DESIGN.md / profiles."""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
import spectro
from spectro.pipeline import stft_pipelined
x = (np.random.default_rng(1234).standard_normal((64, 480000)) * 0.1).astype(np.float32)
xi = (x * 20000).astype(np.int16)
kw = dict(fs=48000.0, nperseg=1024, window="hann", noverlap=768)
spectro.spectrogram(x[:2], **kw)
stft_pipelined(x[:8], **kw)


def bench(name, fn, arg, reps=5):
    best = 1e9
    for rep in range(reps):
        t0 = time.perf_counter()
        f, t, s = fn(arg, **kw)
        dt = time.perf_counter() - t0
        best = min(best, dt)
        frames = s.shape[0] * s.shape[2]
        nbytes = arg.nbytes + s.nbytes
        del s
        gc.collect()
    print(f"{name:58s} {best*1e3:7.2f} ms  {frames/best/1e6:7.1f} Mframes/s  ({nbytes/best/1e9:5.1f} GB/s over PCIe)")


bench("spectro.spectrogram f32 (upload, kernel, download)", spectro.spectrogram, x)
bench("stft_pipelined f32 (2 streams, pinned result, default 16 MB chunks)", stft_pipelined, x)
bench("stft_pipelined int16 in", stft_pipelined, xi)
for mb in (8, 16, 64, 128):
    bench(f"stft_pipelined f32, {mb} MB chunks", lambda a, **k: stft_pipelined(a, chunk_bytes=mb << 20, **k), x, reps=3)
print("floor: 123 MB up + 246 MB down over PCIe Gen5 x16 (63 GB/s per direction, full duplex): max(1.95, 3.9) ms")
