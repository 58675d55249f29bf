#!/usr/bin/env python3
"""Plot latency of the PlotEngine mirror: reference drawing (pcolormesh) against the opt-in image path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
from PlotEngine import PlotEngine
x = (np.random.default_rng(0).standard_normal(480000) * 0.1)
settings = {"nperseg": 1024, "fmin": 0.0, "fmax": 24000.0, "log_scale": True, "mode_raw": "Spectrogram",
            "mode_proc": "None", "draw_raw": True, "draw_proc": False}
for fast in (False, True, False, True):
    eng = PlotEngine()
    t0 = time.perf_counter()
    eng.plot_extra(x, None, 48000.0, dict(settings, fast_image=fast))
    t1 = time.perf_counter()
    eng.fig.canvas.draw()
    t2 = time.perf_counter()
    print(f"fast_image={fast}: plot_extra {1e3*(t1-t0):7.1f} ms + canvas draw {1e3*(t2-t1):7.1f} ms  (image {eng.last_Sxx.shape})")
