"""Pipelined ingest for batch calls: host clips in, host spectra out, with PCIe busy in both directions.

``spectro.spectrogram`` on a batch is upload -> kernel -> download in sequence into a fresh ``np.empty``; for BASELINE cfg2
(123 MB in, 246 MB out) that is 14 ms against a 0.09 ms kernel: 2.2 ms H2D, 4.4 ms D2H and ~7 ms of first-touch page
faults while the DMA engine writes into never-touched pageable pages (docs/LAB_NOTES.md section 5.1).  Here

  * the batch is cut into chunks of whole clips (~16 MB of input each) that alternate between TWO streams, each with its
    own device input / output buffers: chunk i+1's upload overlaps chunk i's download (PCIe is full duplex) and the kernel
    hides behind both;
  * the result lands in PINNED host memory (``sg_host_alloc``) that is handed to the caller as the numpy array itself --
    no page faults under the DMA and no extra copy; the block goes back to a small pool when the array is garbage
    collected, so a loop of batch calls allocates once;
  * int16 PCM goes up as int16 (half the bytes) and is converted by the kernel (``sg_stft_i16``).

Same results as the unpipelined call, bit for bit: the same plan runs on the same samples, only in pieces.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _capi
from .signal import compute_dtype, plan_for, resolve_segments

__all__ = ["stft_pipelined", "pinned_empty", "pinned_pool_clear", "workspace_release"]

_pool_lock = threading.Lock()
_pool: dict[int, list[int]] = {}          # capacity -> free pinned pointers
_POOL_MAX_BYTES = 1 << 30
_pool_bytes = 0


class _PinnedBlock:
    """One ``sg_host_alloc`` block; returns to the pool (or is freed) when the last array over it dies."""

    def __init__(self, nbytes: int):
        global _pool_bytes
        self.capacity = max(int(nbytes), 1)
        with _pool_lock:
            free = _pool.get(self.capacity)
            if free:
                self.ptr = free.pop()
                _pool_bytes -= self.capacity
                return
        p = C.c_void_p()
        _capi.check(_capi.lib().sg_host_alloc(C.byref(p), self.capacity))
        self.ptr = p.value

    def __del__(self):
        global _pool_bytes
        try:
            with _pool_lock:
                if _pool_bytes + self.capacity <= _POOL_MAX_BYTES:
                    _pool.setdefault(self.capacity, []).append(self.ptr)
                    _pool_bytes += self.capacity
                    return
            _capi.lib().sg_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_pool_clear():
    """Free every pooled pinned block (tests, memory-tight callers)."""
    global _pool_bytes
    with _pool_lock:
        for ptrs in _pool.values():
            for p in ptrs:
                _capi.lib().sg_host_free(C.c_void_p(p))
        _pool.clear()
        _pool_bytes = 0


def pinned_empty(shape, dtype):
    """``np.empty`` in pinned host memory; the block lives as long as any view of the returned array."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if len(shape) else 1
    block = _PinnedBlock(n * dt.itemsize)
    raw = (C.c_char * block.capacity).from_address(block.ptr)
    raw._block = block                                      # the ctypes array is the numpy base object: keeps the block alive
    return np.frombuffer(raw, dtype=dt, count=n).reshape(shape)


class _Workspace:
    """Two streams and two (input, output) device buffer pairs kept across calls (grown on demand): a batch loop pays
    hipMalloc / hipFree -- the latter synchronises the device -- and stream creation once, not per call."""

    def __init__(self):
        self.streams, self.d_in, self.d_out, self.device = [], [None, None], [None, None], None

    def get(self, in_bytes, out_bytes):
        dev = _capi.ensure_device()
        if self.device != dev:
            self.release()
            self.device = dev
        L = _capi.lib()
        while len(self.streams) < 2:
            s = C.c_void_p()
            _capi.check(L.sg_stream_create(C.byref(s)))
            self.streams.append(s)
        for k in range(2):
            if self.d_in[k] is None or self.d_in[k].nbytes < in_bytes:
                if self.d_in[k] is not None:
                    self.d_in[k].free(stream=self.streams[k].value)        # last used on stream k (DeviceBuffer's pool rule)
                self.d_in[k] = _capi.DeviceBuffer(in_bytes)
            if self.d_out[k] is None or self.d_out[k].nbytes < out_bytes:
                if self.d_out[k] is not None:
                    self.d_out[k].free(stream=self.streams[k].value)
                self.d_out[k] = _capi.DeviceBuffer(out_bytes)
        return self.streams, self.d_in, self.d_out

    def release(self):
        L = _capi.lib()
        for s in self.streams:
            L.sg_stream_sync(s)                      # the blocks below were last used on these streams
            L.sg_stream_destroy(s)
        for b in self.d_in + self.d_out:
            if b is not None:
                b.free()
        self.streams, self.d_in, self.d_out = [], [None, None], [None, None]


_ws = _Workspace()
_ws_lock = threading.Lock()


def workspace_release():
    """Free the cached streams and device buffers (the pinned pool has its own ``pinned_pool_clear``)."""
    with _ws_lock:
        _ws.release()


def stft_pipelined(x, fs=1.0, window=("tukey", .25), nperseg=None, noverlap=None, nfft=None, detrend="constant",
                   scaling="density", mode="psd", chunk_bytes: int = 16 << 20):
    """``spectrogram(x, ...)`` for ``x[n_clips, n_samples]`` (last axis) with chunked, double-buffered transfers.

    Returns ``(f, t, Sxx[n_clips, n_bins, n_frames])`` exactly as ``spectro.spectrogram`` does; ``Sxx`` is a view of pinned
    host memory.  Modes 'psd', 'magnitude', 'complex', 'angle' (no 'phase': its unwrap is a host pass over the result)."""
    if mode not in ("psd", "magnitude", "complex", "angle"):
        raise ValueError("stft_pipelined: mode must be 'psd', 'magnitude', 'complex' or 'angle'")
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("stft_pipelined takes [n_clips, n_samples]")
    if np.iscomplexobj(x):
        raise NotImplementedError("complex input is outside the device path")
    if detrend not in _capi.DETREND:
        raise ValueError("Trend type must be 'linear' or 'constant'.")
    if scaling not in _capi.SCALING:
        raise ValueError(f"Unknown scaling: {scaling!r}")
    win, nperseg = resolve_segments(window, nperseg, input_length=x.shape[-1])
    noverlap = nperseg // 8 if noverlap is None else int(noverlap)
    nfft = nperseg if nfft is None else int(nfft)
    if nfft < nperseg:
        raise ValueError("nfft must be greater than or equal to nperseg.")
    if noverlap >= nperseg:
        raise ValueError("noverlap must be less than nperseg.")
    hop = nperseg - noverlap
    cdt = compute_dtype(x.dtype)
    code = _capi.F32 if cdt == np.float32 else _capi.F64
    plan = plan_for(win, nperseg, nfft, hop, _capi.DETREND[detrend], fs, _capi.SCALING[scaling], _capi.MODE[mode], code)
    use_i16 = x.dtype == np.int16 and plan.kernel != "bluestein"
    xh = np.ascontiguousarray(x, dtype=np.int16 if use_i16 else cdt)
    n_clips, n_samples = xh.shape
    n_frames, n_bins = plan.n_frames(n_samples), plan.n_bins
    per_bin = 2 if mode == "complex" else 1
    row = n_frames * n_bins * per_bin                        # output elements per clip
    out = pinned_empty((n_clips, n_frames, n_bins * per_bin), cdt)
    f = _capi.freqs(nfft, fs)
    t = _capi.times(n_samples, nperseg, hop, fs)
    if out.size:
        in_clip_bytes = n_samples * xh.dtype.itemsize
        per_chunk = int(max(1, min(n_clips, chunk_bytes // max(in_clip_bytes, 1))))
        L = _capi.lib()
        with _ws_lock:                                        # one pipelined call at a time shares the workspace
            streams, d_in, d_out = _ws.get(per_chunk * in_clip_bytes, per_chunk * row * out.dtype.itemsize)
            n_streams = 2 if n_clips > per_chunk else 1
            try:
                for i, c0 in enumerate(range(0, n_clips, per_chunk)):
                    c1 = min(c0 + per_chunk, n_clips)
                    k = i % n_streams
                    s = streams[k]
                    # same stream => chunk i+2's upload queues behind chunk i's download: the buffers are safe to reuse
                    _capi.check(L.sg_memcpy_h2d(C.c_void_p(d_in[k].ptr), xh[c0:c1].ctypes.data_as(C.c_void_p),
                                                (c1 - c0) * in_clip_bytes, s))
                    plan.stft(d_in[k].ptr, n_samples, n_samples, c1 - c0, d_out[k].ptr, row, stream=s.value, int16=use_i16)
                    _capi.check(L.sg_memcpy_d2h(out[c0:c1].ctypes.data_as(C.c_void_p), C.c_void_p(d_out[k].ptr),
                                                (c1 - c0) * row * out.dtype.itemsize, s))
            finally:
                # also on an error half way: no DMA may still be writing into `out` when its pinned block goes back to the pool
                rcs = [L.sg_stream_sync(s) for s in streams[:n_streams]]
            for rc in rcs:
                _capi.check(rc)
    res = out.view(np.complex64 if cdt == np.float32 else np.complex128) if mode == "complex" else out
    return f, t, np.moveaxis(res.reshape(n_clips, n_frames, n_bins), -1, -2)
