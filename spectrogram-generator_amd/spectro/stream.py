"""Streaming STFT (BASELINE cfg5): multi-channel chunks in, frames out, state resident in HBM.

Not in the reference (it only ever calls the offline function); the contract is offline equivalence: the frames
produced by feeding a signal in arbitrary chunks equal ``spectrogram`` of the whole signal, frame for frame.
Per channel the device keeps a long staging row; a ``feed`` appends the chunk behind the unconsumed tail (ONE 2-D
H2D copy for all channels), runs ONE ``sg_stft`` launch over all channels on ``[tail | chunk]``, fetches the new
frames (one D2H copy) and merely advances the row's start offset.  Only when a row runs out of room is the tail moved
back to its front (two 2-D device copies, once every ~``slack`` chunks): four driver calls per chunk instead of 26.

Round 3, small chunks (``transport="host"``, chosen by the first chunk's size when a feed moves less than 1 MiB -- cfg5 does): the
staging rows and the frame buffer live in PINNED HOST memory and the kernel reads and writes them over PCIe itself, as the
GUI-sized offline calls do (spectro/signal.py): a feed is two host copies, ONE launch and one synchronisation -- no DMA set-up,
no staging through the runtime's pageable-copy path.
"""
from __future__ import annotations

import numpy as np

from . import _capi
from .signal import compute_dtype, plan_for, resolve_segments

__all__ = ["StreamingSTFT"]

_ZC_MAX = 1 << 20          # a feed that moves at most this many bytes runs zero-copy from / to pinned host memory (spectro/signal.py)


class StreamingSTFT:
    def __init__(self, n_channels: int, fs: float, nperseg: int, hop: int | None = None, window=("tukey", .25),
                 detrend="constant", scaling="density", mode="psd", dtype=np.float32, max_chunk: int = 1 << 16,
                 transport: str = "auto"):
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
        if transport not in ("auto", "host", "device"):
            raise ValueError("transport must be 'auto', 'host' or 'device'")
        self._tail_max = (self.nperseg + self.hop + 1) & ~1
        # largest chunk whose feed moves at most _ZC_MAX bytes over PCIe (rows read + frames written by the kernel itself)
        isz = self.dtype.itemsize
        per_sample = self.n_channels * isz * (1.0 + self.n_bins / self.hop)
        room = _ZC_MAX - self.n_channels * isz * (self._tail_max + self.n_bins)
        self._host_chunk = min(self.max_chunk, max(0, int(room / per_sample)) & ~1)
        if transport == "host" and self._host_chunk < 2:
            raise ValueError("transport='host': one frame of this plan already moves more than the zero-copy limit")
        self.transport = None if transport == "auto" else transport     # "auto": the first non-empty chunk decides
        self._buf = self._tmp = self._out = self._rows = self._out_host = None
        self._start = 0                # offset (samples, even) of the first unconsumed sample in every row
        self._fill = 0                 # valid samples per channel behind _start
        self._consumed = 0             # samples dropped from the front so far (= absolute index of _start)
        self.frames_emitted = 0

    _chunk_cap = None

    def _setup(self, transport):
        """Buffers of the chosen transport (allocated at the first feed)."""
        isz = self.dtype.itemsize
        self.transport = transport
        if transport == "host":
            from .pipeline import pinned_empty
            self._chunk_cap = self._host_chunk
            self._slack = 3                                 # moving the tail is a host memmove here: short rows
            self._stride = (self._tail_max + (self._slack + 1) * self._chunk_cap + 1) & ~1
            self._max_frames = (self._tail_max + self._chunk_cap - self.nperseg) // self.hop + 1
            self._rows = pinned_empty((self.n_channels, self._stride), self.dtype)
            self._rows[...] = 0
            self._out_host = pinned_empty((self.n_channels * self._max_frames * self.n_bins,), self.dtype)
            _capi.ensure_device()
        else:
            self._chunk_cap = self.max_chunk
            self._slack = 32                                # chunks appended between two slides of the tail
            self._stride = (self._tail_max + (self._slack + 1) * self.max_chunk + 1) & ~1   # samples per channel row (even: float2 loads stay aligned)
            self._buf = _capi.DeviceBuffer(self.n_channels * self._stride * isz)
            self._tmp = _capi.DeviceBuffer(self.n_channels * self._tail_max * isz)
            self._max_frames = (self._tail_max + self.max_chunk - self.nperseg) // self.hop + 1
            self._out = _capi.DeviceBuffer(self.n_channels * self._max_frames * self.n_bins * isz)

    def _feed_host(self, x, n):
        isz = self.dtype.itemsize
        rows = self._rows
        if self._start + self._fill + n > self._stride:        # out of room: the tail moves to the front of the rows
            rows[:, :self._fill] = rows[:, self._start:self._start + self._fill].copy()
            self._start = 0
        if n:
            rows[:, self._start + self._fill:self._start + self._fill + n] = x
        self._fill += n
        n_frames = self.plan.n_frames(self._fill)
        if n_frames:
            cnt = self.n_channels * n_frames * self.n_bins
            # the kernel reads the pinned rows and writes the pinned frame buffer itself (zero-copy over PCIe)
            self.plan.stft(rows.ctypes.data + self._start * isz, self._fill, self._stride, self.n_channels, self._out_host.ctypes.data,
                           n_frames * self.n_bins)
            _capi.stream_sync()
            out = self._out_host[:cnt].reshape(self.n_channels, n_frames, self.n_bins).copy()
            used = n_frames * self.hop
            self._start += used
            self._fill -= used
            self._consumed += used
        else:
            out = np.empty((self.n_channels, 0, self.n_bins), self.dtype)
        first = self.frames_emitted
        self.frames_emitted += n_frames
        t = (self.nperseg / 2 + (first + np.arange(n_frames)) * float(self.hop)) / self.fs
        return t, np.moveaxis(out, 1, 2)

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
        if self._chunk_cap is None:                            # first feed: choose the transport, allocate its buffers
            if self.transport is None:
                if n == 0:
                    return np.empty(0), np.empty((self.n_channels, self.n_bins, 0), self.dtype)
                self.transport = "host" if 2 <= self._host_chunk and n <= self._host_chunk else "device"
            self._setup(self.transport)
        if n > self._chunk_cap:
            parts = [self.feed(x[:, i:i + self._chunk_cap]) for i in range(0, n, self._chunk_cap)]
            return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts], axis=-1)
        if self.transport == "host":
            return self._feed_host(x, n)
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
            if b is not None:
                b.free()
        self._buf = self._tmp = self._out = self._rows = self._out_host = None
