"""``spectrogram`` -- the swap point of the reference.

The reference imports ``scipy.signal.spectrogram`` at PlotEngine.py:8 and calls it at
PlotEngine.py:113 and :232 as ``spectrogram(data, fs=fs, nperseg=nperseg,
scaling="density", mode="psd")``.  This module provides the same callable (same
keyword set as scipy/signal/_spectral_py.py:816-818, same return shapes / dtypes /
warnings / ValueErrors) running on the MI355X through libspectro.so.

Reference mode = call it the way PlotEngine does: Tukey(0.25), noverlap = nperseg//8,
nfft = nperseg, detrend='constant', one-sided density PSD.  Extended mode = pass
``window=`` / ``noverlap=`` / ``detrend=`` explicitly (Hann / hop 256 for the BASELINE
configs).  There is no CPU path: a missing library or device raises.
"""
from __future__ import annotations

import hashlib
import warnings
from collections import OrderedDict

import numpy as np

from . import _capi
from .windows import get_window

__all__ = ["spectrogram", "resolve_segments", "plan_for", "compute_dtype"]

_MODES = ("psd", "complex", "magnitude", "angle", "phase")
_plan_cache: "OrderedDict[tuple, _capi.Plan]" = OrderedDict()
_PLAN_CACHE_MAX = 32
_window_cache: "OrderedDict[tuple, np.ndarray]" = OrderedDict()      # (window spec, nperseg) -> read-only f64 table
_ft_cache: "OrderedDict[tuple, tuple]" = OrderedDict()               # (n_samples, nperseg, hop, nfft, fs) -> (f, t), read-only
_SMALL_CACHE_MAX = 64
# GUI-sized calls (the reference's own use: one sweep, a few dozen frames) are dominated by the two DMA transfers and
# their synchronisation, not by the kernel: below this many bytes of input + output the kernel reads the samples from, and
# writes the spectrum to, PINNED HOST memory directly (zero-copy over PCIe) -- one launch and one synchronisation.
_ZERO_COPY_MAX_BYTES = int(__import__("os").environ.get("SPECTRO_ZERO_COPY_BYTES", str(1 << 20)))
_staging = {"in": None, "out": None}                                 # pinned numpy byte arrays, grown on demand
_staging_lock = __import__("threading").Lock()                       # one zero-copy call at a time owns the staging pair


def _cached_window(window, nperseg):
    key = (window, int(nperseg))
    w = _window_cache.get(key)
    if w is None:
        w = np.ascontiguousarray(get_window(window, nperseg), np.float64)
        w.setflags(write=False)
        _window_cache[key] = w
        while len(_window_cache) > _SMALL_CACHE_MAX:
            _window_cache.popitem(last=False)
    else:
        _window_cache.move_to_end(key)
    return w


def _freqs_times(n_samples, nperseg, hop, nfft, fs):
    """A7 vectors (bit-exact with scipy, computed by the C side), cached per parameter set; callers get their own copies."""
    key = (int(n_samples), int(nperseg), int(hop), int(nfft), float(fs))
    ft = _ft_cache.get(key)
    if ft is None:
        ft = (_capi.freqs(nfft, fs), _capi.times(n_samples, nperseg, hop, fs))
        _ft_cache[key] = ft
        while len(_ft_cache) > _SMALL_CACHE_MAX:
            _ft_cache.popitem(last=False)
    else:
        _ft_cache.move_to_end(key)
    return ft[0].copy(), ft[1].copy()


def _pinned_bytes(which, nbytes):
    from .pipeline import pinned_empty
    buf = _staging[which]
    if buf is None or buf.nbytes < nbytes:
        buf = pinned_empty((max(int(nbytes), 4096) * 2,), np.uint8)
        _staging[which] = buf
    return buf


def resolve_segments(window, nperseg, input_length):
    """Window / nperseg triage with scipy's rules and messages (scipy:2207-2263)."""
    if isinstance(window, (str, tuple)):
        if nperseg is None:
            nperseg = 256
        if nperseg > input_length:
            warnings.warn(f"nperseg = {nperseg:d} is greater than input length "
                          f" = {input_length:d}, using nperseg = {input_length:d}", stacklevel=3)
            nperseg = input_length
        try:
            win = _cached_window(window, nperseg)
        except TypeError:                                    # unhashable window spec: build it every time
            win = get_window(window, nperseg)
    else:
        win = np.asarray(window)
        if len(win.shape) != 1:
            raise ValueError("window must be 1-D")
        if input_length < win.shape[-1]:
            raise ValueError("window is longer than input signal")
        if nperseg is None:
            nperseg = win.shape[0]
        elif nperseg != win.shape[0]:
            raise ValueError("value specified for nperseg is different from length of window")
    return win, int(nperseg)


def compute_dtype(x_dtype):
    """Arithmetic type of the path: scipy computes in the input's precision (scipy:1976-1981)."""
    return np.float32 if np.result_type(x_dtype, np.complex64) == np.complex64 else np.float64


def plan_for(win, nperseg, nfft, hop, detrend, fs, scaling, mode, dtype):
    """Cached sg_plan for one parameter set (plans own the device window/twiddle tables)."""
    w = np.ascontiguousarray(win, np.float64)
    key = (nperseg, nfft, hop, hashlib.sha1(w.tobytes()).hexdigest(), detrend, float(fs), scaling, mode, dtype,
           _capi.ensure_device())
    p = _plan_cache.get(key)
    if p is None:
        p = _capi.Plan(nperseg, nfft, hop, w, detrend, fs, scaling, mode, dtype)
        _plan_cache[key] = p
        while len(_plan_cache) > _PLAN_CACHE_MAX:
            # drop the entry only: a StreamingSTFT / DeviceSpectrogram / caller of plan_for may still hold the Plan, whose
            # __del__ frees the device tables when the last reference goes
            _plan_cache.popitem(last=False)
    else:
        _plan_cache.move_to_end(key)
    return p


def spectrogram(x, fs=1.0, window=("tukey", .25), nperseg=None, noverlap=None, nfft=None,
                detrend="constant", return_onesided=True, scaling="density", axis=-1, mode="psd"):
    """Drop-in for ``scipy.signal.spectrogram`` (real input, one-sided) on the MI355X.

    Returns ``(f, t, Sxx)``: ``f``/``t`` float64 (bit-exact with scipy), ``Sxx`` with the
    frequency axis where the data axis was and the new time axis last, float32 for
    float32/int16 input and float64 for float64 input (complex for ``mode='complex'``).
    """
    if mode not in _MODES:
        raise ValueError(f"unknown value for mode {mode}, must be one of {list(_MODES)}")
    x = np.asarray(x)
    axis = int(axis)
    win, nperseg = resolve_segments(window, nperseg, input_length=x.shape[axis])
    if noverlap is None:
        noverlap = nperseg // 8                       # scipy:969-970 (spectrogram's own default)
    if np.iscomplexobj(x):
        raise NotImplementedError("complex input (two-sided spectrum) is outside the device path")
    if not return_onesided:
        raise NotImplementedError("return_onesided=False is outside the device path")
    if callable(detrend):
        raise NotImplementedError("callable detrend is outside the device path; use False, 'constant' or 'linear'")
    if detrend not in _capi.DETREND:
        raise ValueError("Trend type must be 'linear' or 'constant'.")
    if scaling not in _capi.SCALING:
        raise ValueError(f"Unknown scaling: {scaling!r}")
    if x.size == 0:
        return np.empty(x.shape), np.empty(x.shape), np.empty(x.shape)
    if nperseg < 1:
        raise ValueError("nperseg must be a positive integer")
    if nfft is None:
        nfft = nperseg
    elif nfft < nperseg:
        raise ValueError("nfft must be greater than or equal to nperseg.")
    nfft = int(nfft)
    noverlap = int(noverlap)
    if noverlap >= nperseg:
        raise ValueError("noverlap must be less than nperseg.")
    hop = nperseg - noverlap

    cdt = compute_dtype(x.dtype)
    if axis != -1:
        x = np.moveaxis(x, axis, -1)
    n_samples = x.shape[-1]
    outer = x.shape[:-1]
    n_clips = int(np.prod(outer)) if outer else 1
    use_i16 = x.dtype == np.int16
    xh = np.ascontiguousarray(x.reshape(n_clips, n_samples), dtype=np.int16 if use_i16 else cdt)

    dev_mode = "angle" if mode == "phase" else mode
    plan = plan_for(win, nperseg, nfft, hop, _capi.DETREND[detrend], fs, _capi.SCALING[scaling],
                    _capi.MODE[dev_mode], _capi.F32 if cdt == np.float32 else _capi.F64)
    if use_i16 and plan.kernel == "bluestein":
        use_i16, xh = False, xh.astype(cdt)
    n_frames, n_bins = plan.n_frames(n_samples), plan.n_bins
    per_bin = 2 if mode == "complex" else 1
    out = np.empty((n_clips, n_frames, n_bins * per_bin), cdt)

    if 0 < xh.nbytes + out.nbytes <= _ZERO_COPY_MAX_BYTES and out.size:
        # zero-copy: the kernel's loads and stores cross PCIe themselves (cfg1: 44.9 -> ~27 us for the device part)
        with _staging_lock:
            s_in, s_out = _pinned_bytes("in", xh.nbytes), _pinned_bytes("out", out.nbytes)
            s_in[:xh.nbytes].view(xh.dtype).reshape(xh.shape)[...] = xh
            plan.stft(s_in.ctypes.data, n_samples, n_samples, n_clips, s_out.ctypes.data, n_frames * n_bins * per_bin, int16=use_i16)
            _capi.stream_sync()
            out[...] = s_out[:out.nbytes].view(out.dtype).reshape(out.shape)
    else:
        d_in = _capi.DeviceBuffer(xh.nbytes)
        d_out = _capi.DeviceBuffer(max(out.nbytes, 8))
        try:
            # (pinning the numpy buffers in place with sg_host_register was measured: 18-27 ms vs 23 ms for the cfg2
            #  batch -- registration costs what it saves; batch callers use spectro.pipeline or keep data on the device)
            d_in.upload(xh)
            plan.stft(d_in.ptr, n_samples, n_samples, n_clips, d_out.ptr, n_frames * n_bins * per_bin, int16=use_i16)
            d_out.download(out)
            _capi.stream_sync()
        finally:
            d_in.free()
            d_out.free()

    if mode == "complex":
        out = out.view(np.complex64 if cdt == np.float32 else np.complex128)
    res = out.reshape(*outer, n_frames, n_bins)
    f, t = _freqs_times(n_samples, nperseg, hop, nfft, fs)
    ax = axis - 1 if axis < 0 else axis
    res = np.moveaxis(res, -1, ax)                     # frequency where the data axis was; time last
    if mode == "phase":
        res = np.unwrap(res, axis=ax)                  # scipy:1003: unwrap along the frequency axis
    return f, t, res
