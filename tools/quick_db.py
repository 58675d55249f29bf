#!/usr/bin/env python3
"""cfg2 log-power leg: fused dB kernel (sg_stft_db) against the linear STFT and against STFT + sg_normalise_image.
python tools/quick_db.py [n_clips] [iters]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spectrogram-generator_amd"))
from spectro import _capi  # noqa: E402
from spectro.windows import get_window  # noqa: E402
import ctypes as C  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N, n, hop = 480000, 1024, 256
_capi.ensure_device()
x = (np.random.default_rng(1234).standard_normal((n_clips, N)) * 0.1).astype(np.float32)
plan = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F32)
nfr = plan.n_frames(N)
NB = 4
ins = [_capi.DeviceBuffer(x.nbytes) for _ in range(NB)]
outs = [_capi.DeviceBuffer(n_clips * nfr * 513 * 4) for _ in range(NB)]
imgs = [_capi.DeviceBuffer(n_clips * nfr * 513 * 4) for _ in range(2)]
mm = _capi.DeviceBuffer(8)
for b in ins:
    b.upload(x)
_capi.stream_sync()
L = _capi.lib()
frames = n_clips * nfr


def lin(i):
    plan.stft(ins[i % NB].ptr, N, N, n_clips, outs[i % NB].ptr, nfr * 513)


def db(i):
    plan.stft_db(ins[i % NB].ptr, N, N, n_clips, 0, 512, 2.5e-6, outs[i % NB].ptr, nfr * 513, mm.ptr)


def db_band(i):
    plan.stft_db(ins[i % NB].ptr, N, N, n_clips, 10, 200, 2.5e-6, outs[i % NB].ptr, nfr * 191, mm.ptr)


def unfused(i):
    plan.stft(ins[i % NB].ptr, N, N, n_clips, outs[i % NB].ptr, nfr * 513)
    _capi.check(L.sg_normalise_image(C.c_void_p(outs[i % NB].ptr), 0, frames, 513, 0, 512, 1, 2.5e-6, C.c_void_p(imgs[i % 2].ptr), C.c_void_p(mm.ptr), None))


def rescale(i):
    _capi.check(L.sg_db_rescale(C.c_void_p(outs[i % NB].ptr), frames * 513, C.c_void_p(mm.ptr), None))


res = {}
for rep in range(2):
    for name, fn in (("linear sg_stft", lin), ("fused sg_stft_db", db), ("fused sg_stft_db band 10..200", db_band),
                     ("sg_stft + sg_normalise_image(log)", unfused), ("sg_db_rescale (in place)", rescale)):
        for i in range(20):
            fn(i)
        _capi.stream_sync()
        t0 = time.perf_counter()
        for i in range(iters):
            fn(i)
        _capi.stream_sync()
        dt = (time.perf_counter() - t0) / iters
        res[name] = min(res.get(name, 1e9), dt)
base = res["linear sg_stft"]
for k, v in res.items():
    print(f"{k:40s} {v*1e6:8.1f} us   x{v/base:5.2f} of linear   {frames/v/1e9:6.3f} G frames/s")
