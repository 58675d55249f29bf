"""Accuracy of the chirp-z kernels next to scipy float32 against the float64 oracle: python tools/acc_np2.py (GPU box)"""
import sys, os
sys.path.insert(0, "spectrogram-generator_amd"); sys.path.insert(0, ".")
import numpy as np
import spectro as sp
from spectro import _capi
from spectro.signal import plan_for
from spectro.windows import get_window
from oracle import stft_oracle as orc
for n, hop, clips, frames in [(2080, 64, 3, 701), (4128, 96, 2, 1031), (2016, 64, 3, 701), (2048, 64, 3, 701), (8192, 64, 2, 641), (4096, 64, 2, 641), (96, 24, 5, 83), (224, 56, 5, 83), (128, 32, 5, 83)]:
    rng = np.random.default_rng(n * 7 + hop)
    ns = n + hop * (frames - 1) + 4                         # (even: an odd clip stride sends the power-of-two register kernels to their LDS fallback)
    x = (rng.standard_normal((clips, ns)) * 0.3 + 0.5).astype(np.float32)
    kw = dict(fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
    plan = plan_for(get_window("hann", n), n, n, hop, 1, 48000.0, 0, 0, _capi.F32)
    _, _, so = orc.spectrogram(x.astype(np.float64), **kw)
    res = {}
    names = [plan.kernel] + (["bluestein"] if (n & (n - 1)) else ["stockham"])
    for k in names:
        plan.force_kernel(k)
        _, _, s = sp.spectrogram(x, **kw)
        fmax = so.max(axis=1, keepdims=True)
        d = np.abs(s - so)
        big = so >= 1e-3 * fmax
        rel = d[big] / so[big]
        print(n, k, "normwise %.2e" % (np.linalg.norm(s - so) / np.linalg.norm(so)), "per-frame %.2e" % (d / fmax).max(), "per-bin rel max %.2e p99.99 %.2e" % (rel.max(), np.quantile(rel, 0.9999)), flush=True)
    plan.force_kernel(names[0])
    import scipy.signal as ss
    _, _, s = ss.spectrogram(x, **kw)
    d = np.abs(s - so); rel = d[big] / so[big]
    print(n, "scipy f32", "normwise %.2e" % (np.linalg.norm(s - so) / np.linalg.norm(so)), "per-frame %.2e" % (d / fmax).max(), "per-bin rel max %.2e" % rel.max(), flush=True)
