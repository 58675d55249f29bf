"""Device-resident spectrogram objects and the epilogues the reference applies to them.

``DeviceSpectrogram`` keeps the frame-major spectrum of one ``stft()`` call in HBM and offers
the reference's post-processing (PlotEngine.py:114-131, :238-241, :686-719) as device kernels:

    mask + store      -> band_slice()        (A8)
    normalise / dB    -> image()             (A9, A10)
    band log-power    -> features()          (A11; fused variant: band_features())
    power summaries   -> total_power(), band_totals()   (A12, A13)

Everything numeric runs through libspectro.so; numpy is used for shapes and host copies only.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _capi
from .signal import compute_dtype, plan_for, resolve_segments

__all__ = ["DeviceSpectrogram", "DeviceClips", "stft", "band_features", "log_image", "bin_range"]


def bin_range(f: np.ndarray, fmin: float, fmax: float):
    """Index range [k_lo, k_hi] of ``(f >= fmin) & (f <= fmax)`` (inclusive both ends,
    PlotEngine.py:114/:238); ``f`` is ascending so the mask is one contiguous run.
    Returns ``(k_lo, k_hi)`` with ``k_lo > k_hi`` when the mask is empty."""
    k_lo = int(np.searchsorted(f, fmin, side="left"))
    k_hi = int(np.searchsorted(f, fmax, side="right")) - 1
    return k_lo, k_hi


def _np_dtype(code):
    return np.float32 if code == _capi.F32 else np.float64


class DeviceSpectrogram:
    """Frame-major PSD ``[n_clips][n_frames][n_bins]`` resident in HBM plus its f / t vectors."""

    def __init__(self, buf, dtype_code, n_clips, n_frames, n_bins, f, t, fs, plan=None, outer=()):
        self.buf, self.dtype_code = buf, dtype_code
        self.n_clips, self.n_frames, self.n_bins = n_clips, n_frames, n_bins
        self.f, self.t, self.fs, self.plan, self.outer = f, t, fs, plan, outer
        self._scratch = _capi.DeviceBuffer(64)

    @property
    def dtype(self):
        return _np_dtype(self.dtype_code)

    @property
    def rows(self):
        return self.n_clips * self.n_frames

    def free(self):
        for b in (self.buf, self._scratch):
            if b is not None:
                b.free()
        self.buf = None

    # ---- host copies -----------------------------------------------------
    def to_host(self):
        """Full spectrum as scipy lays it out: ``[..., n_bins, n_frames]`` (view of frame-major memory)."""
        out = np.empty((self.n_clips, self.n_frames, self.n_bins), self.dtype)
        if out.size:
            self.buf.download(out)
            _capi.stream_sync()
        return np.moveaxis(out.reshape(*self.outer, self.n_frames, self.n_bins), -1, -2)

    def band_slice(self, k_lo, k_hi):
        """A8: ``Sxx[mask, :]`` for one clip batch -> host array ``[..., n_mask, n_frames]``."""
        width = max(k_hi - k_lo + 1, 0)
        out = np.empty((self.n_clips, self.n_frames, width), self.dtype)
        if out.size:
            d = _capi.DeviceBuffer(out.nbytes)
            try:
                _capi.check(_capi.lib().sg_slice_bins(C.c_void_p(self.buf.ptr), self.dtype_code, self.rows, self.n_bins,
                                                      k_lo, k_hi, C.c_void_p(d.ptr), None))
                d.download(out)
                _capi.stream_sync()
            finally:
                d.free()
        return np.moveaxis(out.reshape(*self.outer, self.n_frames, width), -1, -2)

    # ---- A9 / A10 ----------------------------------------------------------
    def image(self, k_lo, k_hi, log_scale, global_max=None):
        """Normalised display image over bins [k_lo, k_hi] (PlotEngine.py:126-131) -> ``[n_mask, n_frames]``.

        ``base = max(S)`` over the band unless ``global_max > 0``; with ``log_scale`` the dB image is
        min-max rescaled to [0, 1] (zeros when the dB range is <= 1e-6)."""
        width = k_hi - k_lo + 1
        out = np.empty((self.n_clips, self.n_frames, width), self.dtype)
        if out.size:
            d = _capi.DeviceBuffer(out.nbytes)
            try:
                gm = 0.0 if (global_max is None or global_max <= 0) else float(global_max)
                _capi.check(_capi.lib().sg_normalise_image(
                    C.c_void_p(self.buf.ptr), self.dtype_code, self.rows, self.n_bins, k_lo, k_hi, int(bool(log_scale)),
                    gm, C.c_void_p(d.ptr), C.c_void_p(self._scratch.ptr), None))
                d.download(out)
                _capi.stream_sync()
            finally:
                d.free()
        return np.moveaxis(out.reshape(*self.outer, self.n_frames, width), -1, -2)

    def image_rgba(self, k_lo, k_hi, log_scale, global_max=None):
        """A9/A10 + colour mapping on the device: the RGBA bytes matplotlib's ``pcolormesh(cmap='jet', vmin=0, vmax=1)``
        would compute from ``image()`` (PlotEngine.py:134-135) -> uint8 ``[n_mask, n_frames, 4]`` ready for ``imshow``."""
        if self.dtype_code != _capi.F32:
            raise ValueError("image_rgba needs an f32 spectrum")
        width = k_hi - k_lo + 1
        n = self.rows * width
        out = np.empty((self.n_clips, self.n_frames, width, 4), np.uint8)
        if n:
            lut = np.empty((256, 4), np.uint8)
            _capi.check(_capi.lib().sg_jet_lut(lut.ctypes.data_as(C.POINTER(C.c_uint8))))
            d_img, d_lut, d_rgba = _capi.DeviceBuffer(n * 4), _capi.DeviceBuffer(1024), _capi.DeviceBuffer(n * 4)
            try:
                d_lut.upload(lut)
                gm = 0.0 if (global_max is None or global_max <= 0) else float(global_max)
                _capi.check(_capi.lib().sg_normalise_image(
                    C.c_void_p(self.buf.ptr), self.dtype_code, self.rows, self.n_bins, k_lo, k_hi, int(bool(log_scale)),
                    gm, C.c_void_p(d_img.ptr), C.c_void_p(self._scratch.ptr), None))
                _capi.check(_capi.lib().sg_colormap(C.c_void_p(d_img.ptr), n, C.c_void_p(d_lut.ptr), C.c_void_p(d_rgba.ptr), None))
                d_rgba.download(out)
                _capi.stream_sync()
            finally:
                for b in (d_img, d_lut, d_rgba):
                    b.free()
        return np.moveaxis(out.reshape(*self.outer, self.n_frames, width, 4), -3, -2)

    def minmax(self, k_lo, k_hi):
        mm = np.empty(2, self.dtype)
        _capi.check(_capi.lib().sg_minmax(C.c_void_p(self.buf.ptr), self.dtype_code, self.rows, self.n_bins, k_lo, k_hi,
                                          C.c_void_p(self._scratch.ptr), None))
        self._scratch.download(mm)
        _capi.stream_sync()
        return mm[0], mm[1]

    # ---- A11 ---------------------------------------------------------------
    def features(self, k_lo, k_hi):
        """``column_stack[log10(sum_band + 1e-20), diff(prepend first)]`` -> ``[n_frames, 2]`` (first clip)."""
        band = _capi.DeviceBuffer(max(self.rows, 1) * np.dtype(self.dtype).itemsize)
        try:
            _capi.check(_capi.lib().sg_band_sum(C.c_void_p(self.buf.ptr), self.dtype_code, self.rows, self.n_bins, k_lo, k_hi,
                                                C.c_void_p(band.ptr), None))
            return _features_from_band(band, self.dtype_code, self.n_clips, self.n_frames)
        finally:
            band.free()

    # ---- A12 / A13 ---------------------------------------------------------
    def band_totals(self, ranges):
        """``sum over frames of sum_{k in [lo, hi)} max(0, S[f][k])`` per half-open bin range (double)."""
        ranges = list(ranges)
        out = np.zeros(len(ranges), np.float64)
        for i in range(0, len(ranges), 16):
            part = ranges[i:i + 16]
            lo = (C.c_int * len(part))(*[int(r[0]) for r in part])
            hi = (C.c_int * len(part))(*[int(r[1]) for r in part])
            d = _capi.DeviceBuffer(8 * len(part))
            try:
                _capi.check(_capi.lib().sg_band_totals(C.c_void_p(self.buf.ptr), self.dtype_code, self.rows, self.n_bins,
                                                       len(part), lo, hi, C.c_void_p(d.ptr), None))
                d.download(out[i:i + len(part)])
                _capi.stream_sync()
            finally:
                d.free()
        return out


class DeviceArray:
    """A result left on the device: an owned ``DeviceBuffer`` plus shape and dtype.  ``torch.as_tensor(arr, device="cuda")`` views it
    without a copy (``__cuda_array_interface__``); keep the object alive for as long as the view is used."""

    def __init__(self, buf, shape, dtype):
        self.buf, self.shape, self.dtype = buf, tuple(int(v) for v in shape), np.dtype(dtype)

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.buf.ptr), False), "version": 2, "strides": None}

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        if out.size:
            self.buf.download(out)
            _capi.stream_sync()
        return out

    def free(self):
        if self.buf is not None:
            self.buf.free()
            self.buf = None


def _features_from_band(band, dtype_code, n_clips, n_frames, keep_on_device=False):
    dt = _np_dtype(dtype_code)
    shape = (n_clips, n_frames, 2)
    nbytes = n_clips * n_frames * 2 * np.dtype(dt).itemsize
    d = _capi.DeviceBuffer(max(nbytes, 8))
    try:
        if nbytes:                    # one launch for the batch; the diff restarts at every clip
            _capi.check(_capi.lib().sg_band_features_batch(C.c_void_p(band.ptr), dtype_code, n_clips, n_frames,
                                                           C.c_void_p(d.ptr), None))
        if keep_on_device:
            _capi.stream_sync()       # the caller may hand the block to another library's stream
            arr, d = DeviceArray(d, shape, dt), None
            return arr
        feats = np.empty(shape, dt)
        if nbytes:
            d.download(feats)
            _capi.stream_sync()
        return feats
    finally:
        if d is not None:
            d.free()


def _prepare(x, fs, window, nperseg, noverlap, nfft, detrend, scaling, mode):
    x = np.asarray(x)
    if np.iscomplexobj(x):
        raise NotImplementedError("complex input is outside the device path")
    win, nperseg = resolve_segments(window, nperseg, input_length=x.shape[-1])
    if noverlap is None:
        noverlap = nperseg // 8
    if nfft is None:
        nfft = nperseg
    elif nfft < nperseg:
        raise ValueError("nfft must be greater than or equal to nperseg.")
    if noverlap >= nperseg:
        raise ValueError("noverlap must be less than nperseg.")
    if detrend not in _capi.DETREND:
        raise ValueError("Trend type must be 'linear' or 'constant'.")
    hop = nperseg - int(noverlap)
    cdt = compute_dtype(x.dtype)
    code = _capi.F32 if cdt == np.float32 else _capi.F64
    plan = plan_for(win, nperseg, int(nfft), hop, _capi.DETREND[detrend], fs, _capi.SCALING[scaling],
                    _capi.MODE[mode], code)
    outer = x.shape[:-1]
    n_clips = int(np.prod(outer)) if outer else 1
    use_i16 = x.dtype == np.int16 and plan.kernel != "bluestein"
    xh = np.ascontiguousarray(x.reshape(n_clips, x.shape[-1]), dtype=np.int16 if use_i16 else cdt)
    return plan, xh, use_i16, outer, n_clips, code, nperseg, hop, int(nfft)


def stft(x, fs=1.0, window=("tukey", .25), nperseg=None, noverlap=None, nfft=None, detrend="constant",
         scaling="density", mode="psd") -> DeviceSpectrogram:
    """Like ``spectro.spectrogram`` (last axis) but the spectrum stays on the device."""
    if mode not in ("psd", "magnitude"):
        raise ValueError("stft() keeps real spectra on the device: mode must be 'psd' or 'magnitude'")
    plan, xh, use_i16, outer, n_clips, code, nperseg, hop, nfft = _prepare(x, fs, window, nperseg, noverlap, nfft, detrend, scaling, mode)
    n_samples = xh.shape[1]
    n_frames, n_bins = plan.n_frames(n_samples), plan.n_bins
    isz = 4 if code == _capi.F32 else 8
    d_in = _capi.DeviceBuffer(max(xh.nbytes, 8))
    d_out = _capi.DeviceBuffer(max(n_clips * n_frames * n_bins * isz, 8))
    try:
        d_in.upload(xh)
        plan.stft(d_in.ptr, n_samples, n_samples, n_clips, d_out.ptr, n_frames * n_bins, int16=use_i16)
        _capi.stream_sync()
    except Exception:
        d_out.free()
        raise
    finally:
        d_in.free()
    return DeviceSpectrogram(d_out, code, n_clips, n_frames, n_bins, _capi.freqs(nfft, fs),
                             _capi.times(n_samples, nperseg, hop, fs), fs, plan, outer)


def band_features(x, fs, nperseg, fmin, fmax, window=("tukey", .25), noverlap=None, detrend="constant"):
    """A11 fused: the spectrogram never reaches HBM -- per frame only the band sum is written (8 B/frame
    instead of 2052 B/frame at nfft = 1024), then log10 / diff.  Returns ``(t, feats[n_frames, 2])`` for a
    1-D signal, ``(None, None)`` when there are no frames (PlotEngine.py:236)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")           # the reference's second call re-triggers the nperseg warning; stay quiet here
        plan, xh, use_i16, outer, n_clips, code, nperseg, hop, nfft = _prepare(x, fs, window, nperseg, noverlap, None, detrend, "density", "psd")
    n_samples = xh.shape[1]
    n_frames = plan.n_frames(n_samples)
    if xh.size == 0 or n_frames == 0:
        return None, None
    from .signal import _freqs_times
    f, t = _freqs_times(n_samples, nperseg, hop, nfft, fs)
    k_lo, k_hi = bin_range(f, fmin, fmax)
    dt = _np_dtype(code)
    if k_lo > k_hi:      # empty mask: np.sum over no rows = 0 -> log10(1e-20) = -20, diff 0 (PlotEngine.py:239-241)
        feats = np.zeros((n_clips, n_frames, 2), dt)
        feats[..., 0] = np.log10(dt(0) + 1e-20)
        return t, feats.reshape(*outer, n_frames, 2)
    if use_i16:
        xh = xh.astype(dt)
    isz = np.dtype(dt).itemsize
    from . import signal as _sig
    if plan.kernel != "bluestein" and xh.nbytes + n_clips * n_frames * 2 * isz <= _sig._ZERO_COPY_MAX_BYTES:
        # GUI-sized call (PlotEngine.py:232 on one sweep): the kernels read the samples from and write the features to pinned
        # host memory directly -- no DMA transfers, one synchronisation (spectro.signal, zero-copy path)
        band = _capi.DeviceBuffer(n_clips * n_frames * isz)
        try:
            with _sig._staging_lock:
                s_in = _sig._pinned_bytes("in", xh.nbytes)
                s_out = _sig._pinned_bytes("out", n_clips * n_frames * 2 * isz)
                s_in[:xh.nbytes].view(xh.dtype).reshape(xh.shape)[...] = xh
                plan.band_power(s_in.ctypes.data, n_samples, n_samples, n_clips, k_lo, k_hi, band.ptr, n_frames)
                _capi.check(_capi.lib().sg_band_features_batch(C.c_void_p(band.ptr), code, n_clips, n_frames, C.c_void_p(s_out.ctypes.data), None))
                _capi.stream_sync()
                feats = s_out[:n_clips * n_frames * 2 * isz].view(dt).reshape(n_clips, n_frames, 2).copy()
        finally:
            band.free()
        return t, feats.reshape(*outer, n_frames, 2)
    d_in = _capi.DeviceBuffer(xh.nbytes)
    band = _capi.DeviceBuffer(n_clips * n_frames * isz)
    try:
        d_in.upload(xh)
        if plan.kernel == "bluestein":
            spec = _capi.DeviceBuffer(n_clips * n_frames * plan.n_bins * isz)
            try:
                plan.stft(d_in.ptr, n_samples, n_samples, n_clips, spec.ptr, n_frames * plan.n_bins)
                _capi.check(_capi.lib().sg_band_sum(C.c_void_p(spec.ptr), code, n_clips * n_frames, plan.n_bins, k_lo, k_hi,
                                                    C.c_void_p(band.ptr), None))
                _capi.stream_sync()
            finally:
                spec.free()
        else:
            plan.band_power(d_in.ptr, n_samples, n_samples, n_clips, k_lo, k_hi, band.ptr, n_frames)
        feats = _features_from_band(band, code, n_clips, n_frames)
    finally:
        d_in.free()
        band.free()
    return t, feats.reshape(*outer, n_frames, 2)


class DeviceClips:
    """A batch of equal-length clips ``[n_clips, n_samples]`` uploaded ONCE and kept in HBM (f32, f64 or int16 PCM), so
    that a parameter sweep (BASELINE cfg4: 15 (n_fft, hop) pairs over the same clips) or several products of one batch
    do not cross PCIe again per call."""

    def __init__(self, x):
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None]
        if x.ndim != 2:
            raise ValueError("DeviceClips takes [n_clips, n_samples]")
        if np.iscomplexobj(x):
            raise NotImplementedError("complex input is outside the device path")
        self.int16 = x.dtype == np.int16
        self.cdt = compute_dtype(x.dtype)
        self.code = _capi.F32 if self.cdt == np.float32 else _capi.F64
        xh = np.ascontiguousarray(x, dtype=np.int16 if self.int16 else self.cdt)
        self.n_clips, self.n_samples = xh.shape
        _capi.ensure_device()
        self.buf = _capi.DeviceBuffer(max(xh.nbytes, 8))
        self.buf.upload(xh)
        _capi.stream_sync()
        self._f = None      # float copy of int16 clips for the kernels that take no int16 (made on first need)

    def free(self):
        for b in (self.buf, self._f):
            if b is not None:
                b.free()
        self.buf = self._f = None

    def _plan(self, fs, window, nperseg, hop, detrend, nfft=None, scaling="density", mode="psd"):
        win, nperseg = resolve_segments(window, nperseg, input_length=self.n_samples)
        if hop is None:
            hop = nperseg - nperseg // 8
        if not (1 <= hop <= nperseg):
            raise ValueError("noverlap must be less than nperseg.")
        if detrend not in _capi.DETREND:
            raise ValueError("Trend type must be 'linear' or 'constant'.")
        nfft = nperseg if nfft is None else int(nfft)
        return plan_for(win, nperseg, nfft, int(hop), _capi.DETREND[detrend], fs, _capi.SCALING[scaling], _capi.MODE[mode],
                        self.code), nperseg, int(hop), nfft

    def _float_ptr(self):
        """device pointer of the clips as the plan's float type (int16 batches: converted once, on the device, when a kernel
        without an int16 front end asks for it)"""
        if not self.int16:
            return self.buf.ptr
        if self._f is None:
            n = self.n_clips * self.n_samples
            self._f = _capi.DeviceBuffer(max(n * 4, 8))
            _capi.check(_capi.lib().sg_convert_i16(C.c_void_p(self.buf.ptr), C.c_void_p(self._f.ptr), n, None))   # on the device
        return self._f.ptr

    def stft(self, fs=1.0, window=("tukey", .25), nperseg=None, hop=None, detrend="constant", scaling="density",
             mode="psd", nfft=None) -> DeviceSpectrogram:
        if mode not in ("psd", "magnitude"):
            raise ValueError("mode must be 'psd' or 'magnitude'")
        plan, nperseg, hop, nfft = self._plan(fs, window, nperseg, hop, detrend, nfft, scaling, mode)
        n_frames, n_bins = plan.n_frames(self.n_samples), plan.n_bins
        isz = np.dtype(self.cdt).itemsize
        out = _capi.DeviceBuffer(max(self.n_clips * n_frames * n_bins * isz, 8))
        i16 = self.int16 and plan.kernel in ("r8x3", "stockham")      # kernels with int16 loads of their own; the others read the float copy
        plan.stft(self.buf.ptr if i16 else self._float_ptr(), self.n_samples, self.n_samples, self.n_clips, out.ptr,
                  n_frames * n_bins, int16=i16)
        return DeviceSpectrogram(out, self.code, self.n_clips, n_frames, n_bins, _capi.freqs(nfft, fs),
                                 _capi.times(self.n_samples, nperseg, hop, fs), fs, plan, (self.n_clips,))

    def band_log_power(self, fs, nperseg, hop, fmin, fmax, window=("tukey", .25), detrend="constant", clip_range=None,
                       keep_on_device=False):
        """A11 for the whole batch (or clips ``clip_range = (a, b)`` of it) in two launches: fused band power (the spectra
        never reach HBM) and log10 / first difference.  -> ``(t, feats[n_clips, n_frames, 2])``; ``(None, None)`` without frames.
        ``keep_on_device``: ``feats`` is a ``DeviceArray`` (the caller frees it) -- what a gather over RCCL wants."""
        plan, nperseg, hop, nfft = self._plan(fs, window, nperseg, hop, detrend)
        n_frames = plan.n_frames(self.n_samples)
        a, b = (0, self.n_clips) if clip_range is None else (int(clip_range[0]), int(clip_range[1]))
        if not (0 <= a <= b <= self.n_clips):
            raise ValueError("clip_range outside the batch")
        if n_frames == 0 or b == a:
            return None, None
        return self._band_log_power(plan, nperseg, hop, nfft, fs, fmin, fmax, n_frames, a, b - a, keep_on_device)

    def _band_log_power(self, plan, nperseg, hop, nfft, fs, fmin, fmax, n_frames, first, count, keep_on_device=False):
        view = _ClipView(self, first, count)
        return view.band_log_power(plan, nperseg, hop, nfft, fs, fmin, fmax, n_frames, keep_on_device)
    def log_image(self, fs, nperseg, hop, fmin, fmax, global_max, window=("tukey", .25), detrend="constant", rescale=True):
        """The log display of PlotEngine.py:126-131 for a caller-supplied ``global_max`` (:110), per batch:
        ``(f_band, t, image[n_clips, n_band, n_frames])`` with the min-max taken over the whole batch.

        nperseg = nfft = 1024 f32 plans run the fused kernel (``sg_stft_db``: the linear spectrum never reaches HBM and
        the extrema come out of the same launch); every other plan composes ``sg_stft`` + ``sg_normalise_image``."""
        if global_max is None or not (global_max > 0):
            raise ValueError("log_image needs the batch-global base: global_max > 0")
        plan, nperseg, hop, nfft = self._plan(fs, window, nperseg, hop, detrend)
        n_frames = plan.n_frames(self.n_samples)
        f = _capi.freqs(nfft, fs)
        t = _capi.times(self.n_samples, nperseg, hop, fs)
        k_lo, k_hi = bin_range(f, fmin, fmax)
        width = max(k_hi - k_lo + 1, 0)
        img = np.empty((self.n_clips, n_frames, width), self.cdt)
        if img.size == 0:
            return f[k_lo:k_hi + 1], t, np.moveaxis(img, -1, -2)
        if plan.kernel == "r8x3" and plan.mode == _capi.MODE["psd"]:
            d_img, mm = _capi.DeviceBuffer(img.nbytes), _capi.DeviceBuffer(8)
            try:
                plan.stft_db(self._float_ptr(), self.n_samples, self.n_samples, self.n_clips, k_lo, k_hi, float(global_max),
                             d_img.ptr, n_frames * width, mm.ptr)
                if rescale:
                    _capi.check(_capi.lib().sg_db_rescale(C.c_void_p(d_img.ptr), img.size, C.c_void_p(mm.ptr), None))
                d_img.download(img)
                _capi.stream_sync()
            finally:
                d_img.free()
                mm.free()
        else:
            if not rescale:
                raise NotImplementedError("the un-rescaled dB image is a product of the fused kernel only")
            dev = self.stft(fs, window, nperseg, hop, detrend)
            try:
                return f[k_lo:k_hi + 1], t, dev.image(k_lo, k_hi, True, global_max)      # [n_clips, n_band, n_frames]
            finally:
                dev.free()
        return f[k_lo:k_hi + 1], t, np.moveaxis(img, -1, -2)


class _ClipView:
    """clips ``first .. first + count`` of a ``DeviceClips`` (a pointer offset: clips are rows of one buffer)"""

    def __init__(self, clips: "DeviceClips", first: int, count: int):
        self.count, self.n_samples, self.cdt, self.code = count, clips.n_samples, clips.cdt, clips.code
        self.ptr = clips._float_ptr() + first * clips.n_samples * np.dtype(clips.cdt).itemsize

    def band_log_power(self, plan, nperseg, hop, nfft, fs, fmin, fmax, n_frames, keep_on_device=False):
        f = _capi.freqs(nfft, fs)
        t = _capi.times(self.n_samples, nperseg, hop, fs)
        k_lo, k_hi = bin_range(f, fmin, fmax)
        if k_lo > k_hi:
            feats = np.zeros((self.count, n_frames, 2), self.cdt)
            feats[..., 0] = np.log10(self.cdt(0) + 1e-20)
            if keep_on_device:
                buf = _capi.DeviceBuffer(max(feats.nbytes, 8))
                buf.upload(feats)
                _capi.stream_sync()
                return t, DeviceArray(buf, feats.shape, feats.dtype)
            return t, feats
        isz = np.dtype(self.cdt).itemsize
        band = _capi.DeviceBuffer(self.count * n_frames * isz)
        try:
            if plan.kernel == "bluestein":
                spec = _capi.DeviceBuffer(self.count * n_frames * plan.n_bins * isz)
                try:
                    plan.stft(self.ptr, self.n_samples, self.n_samples, self.count, spec.ptr, n_frames * plan.n_bins)
                    _capi.check(_capi.lib().sg_band_sum(C.c_void_p(spec.ptr), self.code, self.count * n_frames, plan.n_bins,
                                                        k_lo, k_hi, C.c_void_p(band.ptr), None))
                    _capi.stream_sync()
                finally:
                    spec.free()
            else:
                plan.band_power(self.ptr, self.n_samples, self.n_samples, self.count, k_lo, k_hi, band.ptr, n_frames)
            return t, _features_from_band(band, self.code, self.count, n_frames, keep_on_device)
        finally:
            band.free()


def log_image(x, fs, nperseg, fmin, fmax, global_max, window=("tukey", .25), noverlap=None, detrend="constant"):
    """One-call form of ``DeviceClips.log_image`` for a host array ``[..., n_samples]`` -> ``(f_band, t, image[..., n_band, n_frames])``."""
    x = np.asarray(x)
    outer = x.shape[:-1]
    clips = DeviceClips(x.reshape(-1, x.shape[-1]))
    try:
        win, nperseg = resolve_segments(window, nperseg, input_length=x.shape[-1])
        hop = nperseg - (nperseg // 8 if noverlap is None else int(noverlap))
        f, t, img = clips.log_image(fs, nperseg, hop, fmin, fmax, global_max, window=win, detrend=detrend)
    finally:
        clips.free()
    return f, t, img.reshape(*outer, *img.shape[1:])
