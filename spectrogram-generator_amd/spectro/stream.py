"""Streaming STFT (BASELINE cfg5): multi-channel chunks in, frames out, state resident in HBM.

Not in the reference (it only ever calls the offline function); the contract is offline equivalence: the frames
produced by feeding a signal in arbitrary chunks equal ``spectrogram`` of the whole signal, frame for frame.
Per channel the device keeps the unconsumed tail (< nperseg + hop samples) in front of a staging buffer; every
``feed`` appends the chunk (one H2D copy per call for all channels), runs ONE ``sg_stft`` launch over all channels
on ``[tail | chunk]`` and slides the tail with a device-to-device copy.
"""
from __future__ import annotations

import numpy as np

from . import _capi
from .signal import compute_dtype, plan_for, resolve_segments

__all__ = ["StreamingSTFT"]


class StreamingSTFT:
    def __init__(self, n_channels: int, fs: float, nperseg: int, hop: int | None = None, window=("tukey", .25),
                 detrend="constant", scaling="density", mode="psd", dtype=np.float32, max_chunk: int = 1 << 16):
        if mode not in ("psd", "magnitude"):
            raise ValueError("streaming keeps real spectra: mode must be 'psd' or 'magnitude'")
        self.n_channels, self.fs, self.nperseg = int(n_channels), float(fs), int(nperseg)
        win, _ = resolve_segments(window, self.nperseg, input_length=self.nperseg)
        self.hop = self.nperseg - self.nperseg // 8 if hop is None else int(hop)
        if not (1 <= self.hop <= self.nperseg):
            raise ValueError("noverlap must be less than nperseg.")
        self.dtype = np.dtype(compute_dtype(dtype))
        code = _capi.F32 if self.dtype == np.float32 else _capi.F64
        self.plan = plan_for(win, self.nperseg, self.nperseg, self.hop, _capi.DETREND[detrend], fs,
                             _capi.SCALING[scaling], _capi.MODE[mode], code)
        self.n_bins = self.plan.n_bins
        self.max_chunk = int(max_chunk)
        self._stride = (self.nperseg + self.hop + self.max_chunk + 1) & ~1   # samples per channel (even: keeps float2 loads aligned)
        isz = self.dtype.itemsize
        self._buf = _capi.DeviceBuffer(self.n_channels * self._stride * isz)
        self._tmp = _capi.DeviceBuffer(self.n_channels * (self.nperseg + self.hop) * isz)
        max_frames = (self._stride - self.nperseg) // self.hop + 1
        self._out = _capi.DeviceBuffer(self.n_channels * max_frames * self.n_bins * isz)
        self._max_frames = max_frames
        self._fill = 0                 # valid samples per channel currently in the buffer
        self._consumed = 0             # samples dropped from the front so far (= absolute index of buffer start)
        self.frames_emitted = 0

    def feed(self, chunk):
        """``chunk``: ``[n_channels, n]`` (or 1-D for one channel).  Returns ``(t, S[channel, bin, frame])`` for the
        frames completed by this chunk (possibly zero)."""
        x = np.atleast_2d(np.asarray(chunk))
        if x.shape[0] != self.n_channels:
            raise ValueError(f"expected {self.n_channels} channels, got {x.shape[0]}")
        n = x.shape[1]
        if n > self.max_chunk:
            parts = [self.feed(x[:, i:i + self.max_chunk]) for i in range(0, n, self.max_chunk)]
            return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts], axis=-1)
        isz = self.dtype.itemsize
        lib = _capi.lib()
        import ctypes as C
        xh = np.ascontiguousarray(x, self.dtype)
        keep = []
        for c in range(self.n_channels):           # strided destination: one copy per channel row
            if n:
                keep.append(np.ascontiguousarray(xh[c]))
                _capi.check(lib.sg_memcpy_h2d(C.c_void_p(self._buf.ptr + (c * self._stride + self._fill) * isz),
                                              keep[-1].ctypes.data_as(C.c_void_p), n * isz, None))
        self._fill += n
        n_frames = self.plan.n_frames(self._fill)
        out = np.empty((self.n_channels, n_frames, self.n_bins), self.dtype)
        if n_frames:
            self.plan.stft(self._buf.ptr, self._fill, self._stride, self.n_channels, self._out.ptr, n_frames * self.n_bins)
            self._out.download(out)
            used = n_frames * self.hop             # samples fully consumed; the rest is the next frame's head
            rest = self._fill - used
            if rest:
                for c in range(self.n_channels):   # slide through a scratch buffer (ranges overlap in place)
                    _capi.check(lib.sg_memcpy_d2d(C.c_void_p(self._tmp.ptr + c * (self.nperseg + self.hop) * isz),
                                                  C.c_void_p(self._buf.ptr + (c * self._stride + used) * isz), rest * isz, None))
                for c in range(self.n_channels):
                    _capi.check(lib.sg_memcpy_d2d(C.c_void_p(self._buf.ptr + c * self._stride * isz),
                                                  C.c_void_p(self._tmp.ptr + c * (self.nperseg + self.hop) * isz), rest * isz, None))
            self._fill = rest
            self._consumed += used
        _capi.stream_sync()
        first = self.frames_emitted
        self.frames_emitted += n_frames
        t = (self.nperseg / 2 + (first + np.arange(n_frames)) * float(self.hop)) / self.fs
        return t, np.moveaxis(out, 1, 2)

    def close(self):
        for b in (self._buf, self._tmp, self._out):
            b.free()
