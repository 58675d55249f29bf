"""Mel filterbank epilogue (BASELINE cfg3) -- an ADDITION: the reference has no mel stage.

Definition (this library's own, see include/spectro.h): HTK mel scale, triangular unit-peak filters between
``fmin`` and ``fmax`` at the rfft bin frequencies.  The contraction ``[frames, bins] x [bins, mels]`` runs on the
matrix cores (exact-f32 MFMA) and skips the all-zero blocks of the triangular bank.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi

__all__ = ["MelBank"]


class MelBank:
    """Device-resident filterbank for one (nfft, fs, n_mels, fmin, fmax)."""

    def __init__(self, nfft: int, fs: float, n_mels: int = 80, fmin: float = 0.0, fmax: float | None = None):
        _capi.ensure_device()
        self.nfft, self.fs, self.n_mels = int(nfft), float(fs), int(n_mels)
        self.n_bins = self.nfft // 2 + 1
        self.fmin, self.fmax = float(fmin), float(fs / 2 if fmax is None else fmax)
        w = np.empty((self.n_bins, self.n_mels), np.float64)
        _capi.check(_capi.lib().sg_mel_weights(self.nfft, self.fs, self.n_mels, self.fmin, self.fmax,
                                               w.ctypes.data_as(C.POINTER(C.c_double))))
        self.weights = w
        n_tiles = (self.n_mels + 15) // 16
        self._k_lo, self._k_hi = (C.c_int * n_tiles)(), (C.c_int * n_tiles)()
        _capi.check(_capi.lib().sg_mel_tile_ranges(w.ctypes.data_as(C.POINTER(C.c_double)), self.n_bins, self.n_mels,
                                                   self._k_lo, self._k_hi))
        packed = np.empty((16 * n_tiles, (self.n_bins + 15) // 16 * 16), np.float32)     # transposed + zero padded
        _capi.check(_capi.lib().sg_mel_pack_weights(w.ctypes.data_as(C.POINTER(C.c_double)), self.n_bins, self.n_mels,
                                                    packed.ctypes.data_as(C.POINTER(C.c_float))))
        self._dev = _capi.DeviceBuffer(packed.nbytes)
        self._dev.upload(packed)
        # band-sparse form for the fused kernel (None for a bank that is not band-limited enough: the MFMA kernel takes it)
        self._sparse = None
        ipl = C.c_int(0)
        start, wts = np.zeros(256, np.int32), np.zeros(8 * 256, np.float32)
        first, count = np.zeros(self.n_mels, np.int32), np.zeros(self.n_mels, np.int32)
        rc = _capi.lib().sg_mel_sparse_pack(w.ctypes.data_as(C.POINTER(C.c_double)), self.n_bins, self.n_mels, C.byref(ipl),
                                            start.ctypes.data_as(C.POINTER(C.c_int32)), wts.ctypes.data_as(C.POINTER(C.c_float)),
                                            first.ctypes.data_as(C.POINTER(C.c_int32)), count.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc == _capi.SG_OK:
            n = 64 * ipl.value
            bufs = [_capi.DeviceBuffer(max(a.nbytes, 4)) for a in (start[:n], wts[:8 * n], first, count)]
            for b, a in zip(bufs, (start[:n], wts[:8 * n], first, count)):
                b.upload(np.ascontiguousarray(a))
            self._sparse = (ipl.value, bufs)
        elif rc != _capi.SG_ERR_UNSUPPORTED:
            _capi.check(rc)
        _capi.stream_sync()

    @property
    def tile_ranges(self):
        return list(zip(self._k_lo, self._k_hi))

    def apply_ptr(self, spec_ptr: int, n_rows: int, out_ptr: int, log_scale: bool = False, dense: bool = False, stream=None,
                  kernel=None):
        """``out[n_rows][n_mels] = spec[n_rows][n_bins] x W`` on raw device pointers (f32); asynchronous.
        ``kernel``: None = band-sparse gather when the bank has that form, else the MFMA contraction ("mfma" forces it, as does
        ``dense=True`` or SPECTRO_MEL_MFMA=1)."""
        import os
        if kernel is None:
            kernel = "mfma" if (dense or self._sparse is None or os.environ.get("SPECTRO_MEL_MFMA") == "1") else "sparse"
        if kernel == "sparse":
            ipl, (b_start, b_w, b_first, b_count) = self._sparse
            _capi.check(_capi.lib().sg_mel_sparse(C.c_void_p(spec_ptr), int(n_rows), self.n_bins, C.c_void_p(b_start.ptr), C.c_void_p(b_w.ptr),
                                                  C.c_void_p(b_first.ptr), C.c_void_p(b_count.ptr), ipl, self.n_mels, int(bool(log_scale)),
                                                  C.c_void_p(out_ptr), C.c_void_p(stream)))
            return
        _capi.check(_capi.lib().sg_mel(C.c_void_p(spec_ptr), int(n_rows), self.n_bins, C.c_void_p(self._dev.ptr), self.n_mels,
                                       None if dense else self._k_lo, None if dense else self._k_hi, int(bool(log_scale)),
                                       C.c_void_p(out_ptr), C.c_void_p(stream)))

    def apply(self, dev_spec, log_scale: bool = False, dense: bool = False, kernel=None):
        """Mel spectrogram of a ``DeviceSpectrogram`` (f32) -> host array ``[..., n_mels, n_frames]``."""
        if dev_spec.dtype_code != _capi.F32 or dev_spec.n_bins != self.n_bins:
            raise ValueError("mel needs an f32 spectrum with nfft//2+1 bins of this bank")
        out = np.empty((dev_spec.n_clips, dev_spec.n_frames, self.n_mels), np.float32)
        if out.size:
            d = _capi.DeviceBuffer(out.nbytes)
            try:
                self.apply_ptr(dev_spec.buf.ptr, dev_spec.rows, d.ptr, log_scale, dense, kernel=kernel)
                d.download(out)
                _capi.stream_sync()
            finally:
                d.free()
        return np.moveaxis(out.reshape(*dev_spec.outer, dev_spec.n_frames, self.n_mels), -1, -2)

    def stft_mel(self, x, plan, log_scale: bool = False):
        """Fused STFT -> mel (sg_stft_mel): ``x`` is ``[n_clips, n_samples]`` float32 on the host, ``plan`` an r8x3 PSD plan.
        Returns ``[n_clips, n_mels, n_frames]``; the linear spectrum is never materialised."""
        x = np.ascontiguousarray(np.atleast_2d(x), np.float32)
        n_clips, n_samples = x.shape
        n_frames = plan.n_frames(n_samples)
        out = np.empty((n_clips, n_frames, self.n_mels), np.float32)
        d_in, d_out = _capi.DeviceBuffer(max(x.nbytes, 8)), _capi.DeviceBuffer(max(out.nbytes, 8))
        try:
            d_in.upload(x)
            self.stft_mel_ptr(plan, d_in.ptr, n_samples, n_samples, n_clips, d_out.ptr, n_frames * self.n_mels, log_scale)
            d_out.download(out)
            _capi.stream_sync()
        finally:
            d_in.free()
            d_out.free()
        return np.moveaxis(out, 1, 2)

    def stft_mel_ptr(self, plan, x_ptr, n_samples, clip_stride, n_clips, out_ptr, out_clip_stride, log_scale=False, stream=None,
                     kernel=None):
        """``kernel``: None = the band-sparse epilogue when the bank has one (every triangular bank), else the MFMA tile
        kernel; "sparse" / "mfma" force one (SPECTRO_FUSED_MFMA=1 in the environment forces "mfma" too)."""
        if self.n_bins != plan.n_bins:
            raise ValueError(f"mel bank built for nfft {self.nfft} ({self.n_bins} bins), plan has {plan.n_bins} bins")
        import os
        if kernel is None:
            kernel = "mfma" if (self._sparse is None or os.environ.get("SPECTRO_FUSED_MFMA") == "1") else "sparse"
        if kernel == "sparse":
            if self._sparse is None:
                raise NotImplementedError("this bank has no band-sparse form (more than 256 work items)")
            ipl, (b_start, b_w, b_first, b_count) = self._sparse
            _capi.check(_capi.lib().sg_stft_mel_sparse(plan.handle, C.c_void_p(x_ptr), int(n_samples), int(clip_stride), int(n_clips),
                                                       C.c_void_p(b_start.ptr), C.c_void_p(b_w.ptr), C.c_void_p(b_first.ptr),
                                                       C.c_void_p(b_count.ptr), ipl, self.n_mels, int(bool(log_scale)),
                                                       C.c_void_p(out_ptr), int(out_clip_stride), C.c_void_p(stream)))
            return
        # kernel form of the MFMA tile path: read from the environment HERE (under the GIL), handed over as flags -- the library reads none
        flags = int(bool(log_scale))
        if os.environ.get("SPECTRO_FUSED_WS") == "1":
            flags |= 0x100 | (0x200 if os.environ.get("SPECTRO_FUSED_CONS") == "4" else 0)      # SG_MEL_FORM_WS | SG_MEL_FORM_CONS4
        _capi.check(_capi.lib().sg_stft_mel(plan.handle, C.c_void_p(x_ptr), int(n_samples), int(clip_stride), int(n_clips),
                                            C.c_void_p(self._dev.ptr), self.n_bins, self.n_mels, self._k_lo, self._k_hi,
                                            flags, C.c_void_p(out_ptr), int(out_clip_stride), C.c_void_p(stream)))

    def close(self):
        self._dev.free()
        if self._sparse is not None:
            for b in self._sparse[1]:
                b.free()
            self._sparse = None
