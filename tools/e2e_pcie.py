#!/usr/bin/env python3
"""End-to-end numpy -> numpy rate of spectro.spectrogram on the cfg2 batch (PCIe-inclusive), for DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
import spectro
x = (np.random.default_rng(1234).standard_normal((64, 480000)) * 0.1).astype(np.float32)
spectro.spectrogram(x[:2], fs=48000.0, nperseg=1024, window="hann", noverlap=768)
for rep in range(3):
    t0 = time.perf_counter()
    f, t, s = spectro.spectrogram(x, fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    dt = time.perf_counter() - t0
    print(f"spectrogram(64x480000 f32) numpy->numpy: {dt*1e3:.1f} ms  {s.shape[0]*s.shape[2]/dt/1e6:.1f} Mframes/s  ({(x.nbytes+s.nbytes)/dt/1e9:.1f} GB/s over PCIe)")
