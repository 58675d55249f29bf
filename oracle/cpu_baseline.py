"""CPU baseline leg of bench.py -- TEST/BENCH INFRASTRUCTURE, never imported by the product.

Times what the reference itself runs on this path: ``scipy.signal.spectrogram`` (the callable
imported at PlotEngine.py:8 and invoked at PlotEngine.py:113/232) with the benchmark's argument
set, on the host cores of the box bench.py runs on.  If scipy is absent there, the numpy
restatement (oracle/stft_oracle.py) is timed instead and the result is labelled "port".
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _callable():
    try:
        import scipy
        from scipy.signal import spectrogram
        return spectrogram, "reference", f"scipy {scipy.__version__}"
    except Exception:                                   # pragma: no cover - scipy is in the image
        from oracle.stft_oracle import spectrogram
        return spectrogram, "port", f"numpy {np.__version__} restatement"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_mode(fn, clips, kw, n_frames, budget_s, threads):
    """best-of-N wall time of ``fn(clip, **kw)`` over all clips: one thread, then a clip-parallel pool"""
    def one(c):
        return fn(clips[c], **kw)[2].shape

    def run_serial():
        t0 = time.perf_counter()
        for c in range(clips.shape[0]):
            one(c)
        return time.perf_counter() - t0

    def run_pool(pool):
        t0 = time.perf_counter()
        list(pool.map(one, range(clips.shape[0])))
        return time.perf_counter() - t0

    one(0)                                              # warm-up (plan caches, page faults)
    t_start = time.perf_counter()
    best1 = run_serial()
    reps1 = 1
    while time.perf_counter() - t_start < budget_s * 0.5 and reps1 < 5:
        best1 = min(best1, run_serial())
        reps1 += 1
    with ThreadPoolExecutor(threads) as pool:
        run_pool(pool)
        bestn = run_pool(pool)
        repsn = 1
        while time.perf_counter() - t_start < budget_s and repsn < 5:
            bestn = min(bestn, run_pool(pool))
            repsn += 1
    return n_frames / best1, n_frames / bestn, reps1, repsn


def time_cpu_baseline(clips: np.ndarray, fs: float, nperseg: int, hop: int, window="hann",
                      budget_s: float = 20.0, threads: int | None = None):
    """Best-of-N wall time over ``clips`` ([n_clips, N] f32), single thread and a clip-parallel pool, for

      * the benchmark's own argument set (BASELINE "extended" mode: explicit Hann window and hop), and
      * the reference's literal call ``spectrogram(x, fs=fs, nperseg=nperseg, scaling="density", mode="psd")``
        (PlotEngine.py:113: Tukey(0.25) window, hop = nperseg - nperseg//8) -> ``reference_mode``.

    Returns a dict ready to be dropped into bench.py's ``cpu_baseline`` object (``value`` = extended mode, pool).
    """
    fn, kind, lib = _callable()
    threads = threads or min(os.cpu_count() or 1, clips.shape[0])
    kw = dict(fs=fs, nperseg=nperseg, window=window, noverlap=nperseg - hop, scaling="density", mode="psd")
    n_frames = ((clips.shape[1] - nperseg) // hop + 1) * clips.shape[0]
    v1, vn, reps1, repsn = _time_mode(fn, clips, kw, n_frames, budget_s * 0.65, threads)
    ref_hop = nperseg - nperseg // 8
    kw_ref = dict(fs=fs, nperseg=nperseg, scaling="density", mode="psd")
    n_frames_ref = ((clips.shape[1] - nperseg) // ref_hop + 1) * clips.shape[0]
    r1, rn, rreps1, rrepsn = _time_mode(fn, clips, kw_ref, n_frames_ref, budget_s * 0.25, threads)
    # ... and on float64 clips, the dtype the reference's loaders hand over (SweepManager.py:135-136): scipy computes in it
    d1, dn, _, dreps = _time_mode(fn, clips.astype(np.float64), kw_ref, n_frames_ref, budget_s * 0.10, threads)
    return {
        "value": vn, "unit": "frames/s", "cores": threads, "kind": kind,
        "single_thread_value": v1,
        "sample": (f"{lib} spectrogram(fs={fs:g}, nperseg={nperseg}, window='{window}', noverlap={nperseg - hop}, "
                   f"density psd) on {clips.shape[0]} clips x {clips.shape[1]} f32 samples = {n_frames} frames; "
                   f"best of {repsn} (pool of {threads} threads over clips) / best of {reps1} (1 thread); "
                   f"host: {cpu_model()}, {os.cpu_count()} logical CPUs"),
        "reference_mode": {
            "value": rn, "single_thread_value": r1, "unit": "frames/s", "cores": threads, "frames": n_frames_ref,
            "f64_value": dn, "f64_single_thread_value": d1,
            "sample": (f"the reference's literal call (PlotEngine.py:113): spectrogram(x, fs={fs:g}, nperseg={nperseg}, "
                       f"scaling='density', mode='psd') -> Tukey(0.25), hop {ref_hop}; same clips = {n_frames_ref} frames; "
                       f"best of {rrepsn} (pool) / best of {rreps1} (1 thread); f64_value: the same call on float64 copies, best of {dreps}"),
        },
    }
