"""Streaming STFT (BASELINE cfg5): multi-channel chunks in, frames out, state resident in HBM.

Not in the reference (it only ever calls the offline function); the contract is offline equivalence: the frames
produced by feeding a signal in arbitrary chunks equal ``spectrogram`` of the whole signal, frame for frame.
Per channel the device keeps a long staging row; a ``feed`` appends the chunk behind the unconsumed tail (ONE 2-D
H2D copy for all channels), runs ONE ``sg_stft`` launch over all channels on ``[tail | chunk]``, fetches the new
frames (one D2H copy) and merely advances the row's start offset.  Only when a row runs out of room is the tail moved
back to its front (two 2-D device copies, once every ~``slack`` chunks): four driver calls per chunk instead of 26.
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
        self._slack = 32                                # chunks appended between two slides of the tail
        self._tail_max = (self.nperseg + self.hop + 1) & ~1
        self._stride = (self._tail_max + (self._slack + 1) * self.max_chunk + 1) & ~1   # samples per channel row (even: float2 loads stay aligned)
        isz = self.dtype.itemsize
        self._buf = _capi.DeviceBuffer(self.n_channels * self._stride * isz)
        self._tmp = _capi.DeviceBuffer(self.n_channels * self._tail_max * isz)
        max_frames = (self._tail_max + self.max_chunk - self.nperseg) // self.hop + 1
        self._out = _capi.DeviceBuffer(self.n_channels * max_frames * self.n_bins * isz)
        self._max_frames = max_frames
        self._start = 0                # offset (samples, even) of the first unconsumed sample in every row
        self._fill = 0                 # valid samples per channel behind _start
        self._consumed = 0             # samples dropped from the front so far (= absolute index of _start)
        self.frames_emitted = 0

    def _copy2d(self, dst, dst_pitch, src, src_pitch, width, kind):
        import ctypes as C
        _capi.check(_capi.lib().sg_memcpy2d(C.c_void_p(dst), dst_pitch, C.c_void_p(src), src_pitch, width, self.n_channels, kind, None))

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
        row = self._stride * isz
        if self._start + self._fill + n > self._stride:        # out of room: move the tail to the front of the rows
            if self._fill:                                     # (through a scratch buffer: the ranges may overlap)
                self._copy2d(self._tmp.ptr, self._tail_max * isz, self._buf.ptr + self._start * isz, row, self._fill * isz, 2)
                self._copy2d(self._buf.ptr, row, self._tmp.ptr, self._tail_max * isz, self._fill * isz, 2)
            self._start = 0
        xh = np.ascontiguousarray(x, self.dtype)
        if n:
            self._copy2d(self._buf.ptr + (self._start + self._fill) * isz, row, xh.ctypes.data, n * isz, n * isz, 0)
        self._fill += n
        n_frames = self.plan.n_frames(self._fill)
        out = np.empty((self.n_channels, n_frames, self.n_bins), self.dtype)
        if n_frames:
            self.plan.stft(self._buf.ptr + self._start * isz, self._fill, self._stride, self.n_channels, self._out.ptr,
                           n_frames * self.n_bins)
            self._out.download(out)
            used = n_frames * self.hop             # samples fully consumed; the rest is the next frame's head
            self._start += used                    # (an even hop keeps the rows 8-byte aligned for the register kernels)
            self._fill -= used
            self._consumed += used
        _capi.stream_sync()
        first = self.frames_emitted
        self.frames_emitted += n_frames
        t = (self.nperseg / 2 + (first + np.arange(n_frames)) * float(self.hop)) / self.fs
        return t, np.moveaxis(out, 1, 2)

    def close(self):
        for b in (self._buf, self._tmp, self._out):
            b.free()
