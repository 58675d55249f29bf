#!/usr/bin/env python3
"""One-off fuzz of the device path against the oracle (on the GPU box): batches whose persistent-kernel runs cross many
clip boundaries, every register-kernel family, modes / scalings / padding / int16.   python tools/fuzz_gpu.py [cases] [seed]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
import spectro
from oracle import stft_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
fam = {}
for it in range(cases):
    nper = int(rng.choice([64, 128, 256, 256, 512, 512, 1024, 1024, 2048, 2048, 4096, 4096, 1000, 384]))
    r = rng.random()
    hop = int(rng.choice([64, 128, 256, 512])) if r < 0.5 else (nper - nper // 8 if r < 0.7 else int(rng.integers(1, nper + 1)))
    hop = max(1, min(hop, nper))
    n_clips = int(rng.choice([1, 2, 3, 7, 16, 33, 70]))
    n_frames = int(rng.choice([1, 2, 3, 5, 17, 64, 131, 400]))
    N = nper + hop * (n_frames - 1) + int(rng.integers(0, hop))
    mode = str(rng.choice(["psd", "psd", "psd", "magnitude", "complex", "angle"]))
    scaling = str(rng.choice(["density", "spectrum"]))
    detrend = [False, "constant", "constant", "linear"][int(rng.integers(0, 4))]
    nfft = nper if rng.random() < 0.8 else int(2 ** np.ceil(np.log2(nper)) * rng.choice([1, 2]))
    nfft = max(nfft, nper)
    dt = rng.choice(["f32", "f32", "f32", "f64", "i16"])
    x = rng.standard_normal((n_clips, N)) * rng.uniform(0.01, 3.0) + rng.uniform(-1, 1)
    x = (x * 3000).astype(np.int16) if dt == "i16" else x.astype(np.float32 if dt == "f32" else np.float64)
    kw = dict(fs=float(rng.choice([500.0, 16000.0, 48000.0])), nperseg=nper, noverlap=nper - hop, nfft=nfft,
              window=[("tukey", 0.25), "hann", "boxcar"][int(rng.integers(0, 3))], detrend=detrend, scaling=scaling, mode=mode)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f, t, s = spectro.spectrogram(x, **kw)
            fo, to, so = orc.spectrogram(x, **kw)
        ok = np.array_equal(f, fo) and np.array_equal(t, to) and s.shape == so.shape and s.dtype == so.dtype
        if ok and s.size:
            if mode == "angle":
                # phase of weak bins is ill-conditioned: compare where the magnitude is not tiny
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    mag = np.abs(orc.spectrogram(x, **{**kw, "mode": "complex"})[2])
                strong = mag > 1e-3 * mag.max(axis=-2, keepdims=True)
                d = np.abs(np.angle(np.exp(1j * (s - so))))
                ok = bool(np.all(d[strong] < (2e-2 if so.dtype == np.float32 else 1e-6)))
            else:
                ref = np.abs(so).max(axis=-2, keepdims=True)
                tol = 2e-4 if np.abs(so).dtype == np.float32 or so.dtype == np.complex64 else 1e-9
                ok = bool(np.all(np.abs(s - so) <= tol * ref + 1e-30))
    except Exception as e:          # noqa: BLE001
        ok = False
        print("EXC", repr(e))
    if not ok:
        bad += 1
        print("FAIL", it, dict(n_clips=n_clips, N=N, dt=str(dt), **kw))
print(f"{cases - bad}/{cases} cases agree with the oracle")
sys.exit(1 if bad else 0)
