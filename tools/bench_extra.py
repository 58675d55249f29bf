#!/usr/bin/env python3
"""Secondary measurements (not the headline bench): cfg3 mel epilogue, cfg4 parameter sweep, cfg5 streaming.
Prints one JSON object; run on the GPU box:  python tools/bench_extra.py > gpurun_out/extra.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
from spectro import _capi  # noqa: E402
from spectro.mel import MelBank  # noqa: E402
from spectro.stream import StreamingSTFT  # noqa: E402
from spectro.windows import get_window  # noqa: E402

_capi.ensure_device()
res = {"device": _capi.device_info()}
rng = np.random.default_rng(1234)
N, n_clips = 480000, 64
x = (rng.standard_normal((n_clips, N)) * 0.1).astype(np.float32)
d_in = _capi.DeviceBuffer(x.nbytes)
d_in.upload(x)


def timed(fn, iters=20, warm=3, settle_s=0.0):
    """average seconds per call; settle_s > 0: that long of back-to-back calls first (clocks / board power as a batch job
    holds them -- a 3-call warm-up times the idle clock)"""
    for _ in range(warm):
        fn()
    _capi.stream_sync()
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < settle_s:
        for _ in range(16):
            fn()
        _capi.stream_sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    _capi.stream_sync()
    return (time.perf_counter() - t0) / iters


# ---- cfg3: STFT + 80-band mel (separate epilogue kernel over the f32 PSD) ----
plan = _capi.Plan(1024, 1024, 256, get_window("hann", 1024), 1, 48000.0, 0, 0, _capi.F32)
nfr = plan.n_frames(N)
d_spec = _capi.DeviceBuffer(n_clips * nfr * 513 * 4)
d_mel = _capi.DeviceBuffer(n_clips * nfr * 80 * 4)
bank = MelBank(1024, 48000.0, 80, 0.0, 24000.0)
plan.stft(d_in.ptr, N, N, n_clips, d_spec.ptr, nfr * 513)
frames = n_clips * nfr
t_stft = timed(lambda: plan.stft(d_in.ptr, N, N, n_clips, d_spec.ptr, nfr * 513), iters=200, settle_s=0.5)
t_mel = timed(lambda: bank.apply_ptr(d_spec.ptr, frames, d_mel.ptr, True, kernel="mfma"), iters=200, settle_s=0.5)
t_mel_sparse = timed(lambda: bank.apply_ptr(d_spec.ptr, frames, d_mel.ptr, True), iters=200, settle_s=0.5)
t_mel_dense = timed(lambda: bank.apply_ptr(d_spec.ptr, frames, d_mel.ptr, True, dense=True), iters=100, settle_s=0.3)
t_fused_mfma = timed(lambda: bank.stft_mel_ptr(plan, d_in.ptr, N, N, n_clips, d_mel.ptr, nfr * 80, True, kernel="mfma"), iters=200, settle_s=0.5)
t_fused = timed(lambda: bank.stft_mel_ptr(plan, d_in.ptr, N, N, n_clips, d_mel.ptr, nfr * 80, True), iters=200, settle_s=0.5)
steps_sparse = sum((hi - lo) // 4 for lo, hi in bank.tile_ranges)
flops_dense = 2.0 * 513 * 80 * frames
res["cfg3_mel"] = {
    "frames": frames, "stft_us": t_stft * 1e6, "mel_us": t_mel * 1e6, "mel_dense_us": t_mel_dense * 1e6,
    "mel_band_sparse_us": t_mel_sparse * 1e6, "mel_band_sparse_hbm_GBps": frames * (513 + 80) * 4 / t_mel_sparse / 1e9,
    "stft_plus_mel_frames_per_s": frames / (t_stft + t_mel),
    "fused_us": t_fused * 1e6, "fused_frames_per_s": frames / t_fused,
    "fused_kernel": "stft1024_r8x3_kernel OUT_MEL (band-sparse epilogue, %d work items per lane)" % (bank._sparse[0] if bank._sparse else 0),
    "fused_algorithmic_GBps": frames * (256 + 80) * 4 / t_fused / 1e9,
    "fused_mfma_tile_kernel_us": t_fused_mfma * 1e6,
    "fused_mfma_issued_TFLOPs": frames / 16 * steps_sparse * 2 * 16 * 16 * 4 / t_fused_mfma / 1e12,
    "mel_hbm_GBps": frames * (513 + 80) * 4 / t_mel / 1e9,
    "mfma_issued_TFLOPs_sparse": frames / 16 * steps_sparse * 2 * 16 * 16 * 4 / t_mel / 1e12,
    "mfma_issued_TFLOPs_dense": frames / 16 * 5 * 129 * 2 * 16 * 16 * 4 / t_mel_dense / 1e12,
    "mfma_f32_peak_TFLOPs": 157.3, "useful_dense_TFLOPs": flops_dense / t_mel / 1e12,
    "note": "exact-f32 MFMA 16x16x4; block-sparse skips all-zero 16-mel x 4-bin blocks of the triangular bank",
}

# ---- cfg4: parameter sweep n_fft x hop on the 64 resident clips (single GPU share of the 256-clip job) ----
# four rotating input/output sets per shape, like bench.py: a single 123 + 124 MB pair would sit in the 256 MiB Infinity Cache
NB = 4
ins = [d_in] + [_capi.DeviceBuffer(x.nbytes) for _ in range(NB - 1)]
for b in ins[1:]:
    b.upload(x)
_capi.stream_sync()
sweep = {}
for n in (256, 512, 1024, 2048, 4096):
    for hop in (64, 128, 256):
        p = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F32)
        nf = p.n_frames(N)
        outs = [_capi.DeviceBuffer(n_clips * nf * (n // 2 + 1) * 4) for _ in range(NB)]
        turn = [0]

        def step():
            i = turn[0] % NB
            turn[0] += 1
            p.stft(ins[i].ptr, N, N, n_clips, outs[i].ptr, nf * (n // 2 + 1))
        t = timed(step, iters=max(8, int(0.05 / max(1e-5, 1e-9 * n_clips * nf * (n / 256)))), warm=4, settle_s=0.25)
        bpf = hop * 4 + (n // 2 + 1) * 4
        sweep[f"n{n}_h{hop}"] = {"kernel": p.kernel, "frames": n_clips * nf, "ms": t * 1e3, "frames_per_s": n_clips * nf / t,
                                 "algorithmic_GBps": n_clips * nf * bpf / t / 1e9}
        for o in outs:
            o.free()
        p.close()
res["cfg4_sweep_64clips"] = sweep
res["cfg4_total_ms"] = sum(v["ms"] for v in sweep.values())
# hops 64 / 128 / 256 of one n_fft share the hop-64 transform (frame i at hop h IS frame i*h/64 at hop 64: spectro.sweep.hop_families);
# a consumer that takes the coarser hops as row subsets of the hop-64 spectrum pays 5 transforms instead of 15
res["cfg4_total_shared_hops_ms"] = sum(v["ms"] for k, v in sweep.items() if k.endswith("_h64"))

# the sweep's REDUCED product (what spectro.sweep.sharded_sweep moves: per-frame band power, fused -- no spectrum is written):
# 15 fused launches, and the 5 launches hop sharing leaves
band = {}
d_bp = _capi.DeviceBuffer(n_clips * ((N - 256) // 64 + 1) * 4)
for n in (256, 512, 1024, 2048, 4096):
    for hop in (64, 128, 256):
        p = _capi.Plan(n, n, hop, get_window("hann", n), 1, 48000.0, 0, 0, _capi.F32)
        nf = p.n_frames(N)
        turn = [0]

        def step():
            turn[0] += 1
            p.band_power(ins[turn[0] % NB].ptr, N, N, n_clips, 1, n // 4, d_bp.ptr, nf)
        t = timed(step, iters=max(8, int(0.05 / max(1e-5, 1e-9 * n_clips * nf * (n / 256)))), warm=4, settle_s=0.25)
        band[f"n{n}_h{hop}"] = {"kernel": p.kernel, "frames": n_clips * nf, "ms": t * 1e3, "frames_per_s": n_clips * nf / t}
        p.close()
d_bp.free()
res["cfg4_band_power_sweep_64clips"] = band
res["cfg4_band_power_total_ms"] = sum(v["ms"] for v in band.values())
res["cfg4_band_power_total_shared_hops_ms"] = sum(v["ms"] for k, v in band.items() if k.endswith("_h64"))

# ---- cfg5: streaming 8 ch x 96 kHz, n_fft 4096 hop 1024, 4096-sample chunks ----
st = StreamingSTFT(8, 96000.0, 4096, 1024, window="hann")
chunk = (rng.standard_normal((8, 4096)) * 0.1).astype(np.float32)
for _ in range(5):
    st.feed(chunk)
t0 = time.perf_counter()
n_chunks, got = 200, 0
lat = []
for _ in range(n_chunks):
    c0 = time.perf_counter()
    t, s = st.feed(chunk)
    lat.append(time.perf_counter() - c0)
    got += s.shape[-1] * 8
dt = time.perf_counter() - t0
res["cfg5_streaming"] = {"chunks": n_chunks, "frames": got, "frames_per_s": got / dt,
                         "realtime_factor": (n_chunks * 4096 / 96000.0) / dt,
                         "chunk_latency_ms_median": float(np.median(lat) * 1e3), "chunk_latency_ms_p99": float(np.percentile(lat, 99) * 1e3),
                         "note": "synchronous feed(), host to host (PCIe-inclusive); chunks this small run with their rows and frames in pinned host memory, the kernel crossing PCIe itself"}
print(json.dumps(res, indent=1))
