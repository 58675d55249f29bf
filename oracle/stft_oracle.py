"""CPU oracle for the STFT/PSD hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the algorithm the reference runs on
its hot path.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``spectro``) never imports
anything from ``oracle/`` and fails loudly when the HIP library is missing.

Where the algorithm lives
-------------------------
The reference's STFT is one third-party call made at ``PlotEngine.py:113`` and
``PlotEngine.py:232``::

    spectrogram(data, fs=fs, nperseg=nperseg, scaling="density", mode="psd")

``spectrogram`` is ``scipy.signal.spectrogram`` (``PlotEngine.py:8``); scipy is
NOT vendored under /root/reference and the reference pins no version.  We pin
it to **scipy 1.15.3 / numpy 2.2.6** (the versions in the build container) and
restate the published algorithm of ``scipy/signal/_spectral_py.py``
(``spectrogram`` l.816, ``_spectral_helper`` l.1863, ``_fft_helper`` l.2158,
``_triage_segments`` l.2207) below.  Citations of the form ``scipy:NNN`` refer
to that file; ``PlotEngine.py:NNN`` / ``SweepManager.py:NNN`` to the reference.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against the
golden vectors in ``tests/golden/*.npz``; those were produced by
``tests/golden/make_golden.py`` from (a) scipy 1.15.3 called with the
reference's exact argument set and (b) the reference's own
``PlotEngine._plot_spectrogram`` / ``_calculate_features`` /
``calculate_*_power`` imported in the build container.  The reference itself
ships no tests, fixtures or golden vectors for this path.
"""
from __future__ import annotations

import warnings

import numpy as np

__all__ = [
    "tukey_periodic", "hann_periodic", "resolve_window", "frame_count",
    "freq_vector", "time_vector", "spectrogram", "band_mask", "plot_image",
    "hmm_features", "absolute_power", "band_powers", "get_signal",
    "EEG_BANDS",
]


# --------------------------------------------------------------------------
# A1 -- window tables (scipy.signal.get_window(..., fftbins=True))
# --------------------------------------------------------------------------
def tukey_periodic(n: int, alpha: float = 0.25) -> np.ndarray:
    """Periodic Tukey window, f64: ``tukey(n + 1, alpha)[:-1]``.

    Follows scipy/signal/windows/_windows.py:866-888 with ``sym=False``
    (``_extend`` adds one point, ``_truncate`` drops it again).
    """
    if n <= 1:
        return np.ones(n)
    if alpha <= 0:
        return np.ones(n, "d")
    if alpha >= 1.0:
        return hann_periodic(n)
    m = n + 1
    k = np.arange(0, m)
    width = int(np.floor(alpha * (m - 1) / 2.0))
    k1 = k[0:width + 1]
    k2 = k[width + 1:m - width - 1]
    k3 = k[m - width - 1:]
    w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * k1 / alpha / (m - 1))))
    w2 = np.ones(k2.shape)
    w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * k3 / alpha / (m - 1))))
    return np.concatenate((w1, w2, w3))[:-1]


def hann_periodic(n: int) -> np.ndarray:
    """Periodic Hann window, f64 (general_cosine with a=[0.5, 0.5], sym=False)."""
    if n <= 1:
        return np.ones(n)
    m = n + 1
    fac = np.linspace(-np.pi, np.pi, m)
    w = np.zeros(m)
    for k, a in enumerate([0.5, 0.5]):
        w += a * np.cos(k * fac)
    return w[:-1]


def resolve_window(window, nperseg, input_length):
    """A1: ``_triage_segments`` (scipy:2207-2263).

    Returns ``(win_f64, nperseg)``.  ``window`` is ``('tukey', a)``, ``'hann'``,
    ``'boxcar'`` or an array.  ``nperseg > input_length`` clamps with a
    ``UserWarning`` exactly like scipy (scipy:2245-2249).
    """
    if isinstance(window, (str, tuple)):
        if nperseg is None:
            nperseg = 256
        if nperseg > input_length:
            warnings.warn(f"nperseg = {nperseg:d} is greater than input length "
                          f" = {input_length:d}, using nperseg = {input_length:d}",
                          stacklevel=3)
            nperseg = input_length
        if isinstance(window, tuple):
            name, args = window[0], window[1:]
        else:
            name, args = window, ()
        if name == "tukey":
            win = tukey_periodic(nperseg, *(args or (0.5,)))
        elif name in ("hann", "hanning"):
            win = hann_periodic(nperseg)
        elif name in ("boxcar", "ones", "rect", "rectangular"):
            win = np.ones(nperseg)
        else:
            raise ValueError(f"oracle restates tukey/hann/boxcar only, got {name!r}")
    else:
        win = np.asarray(window)
        if win.ndim != 1:
            raise ValueError("window must be 1-D")
        if input_length < win.shape[-1]:
            raise ValueError("window is longer than input signal")
        if nperseg is None:
            nperseg = win.shape[0]
        elif nperseg != win.shape[0]:
            raise ValueError("value specified for nperseg is different from length of window")
    return win, int(nperseg)


# --------------------------------------------------------------------------
# A2 / A7 -- integer framing and the f / t vectors (bit-exact targets)
# --------------------------------------------------------------------------
def frame_count(n_samples: int, nperseg: int, step: int) -> int:
    """A2: frames of ``sliding_window_view(x, nperseg)[::step]`` (scipy:2180-2188).

    No padding, no centring, tail dropped: ``(N - n) // step + 1``.
    """
    if n_samples < nperseg:
        return 0
    return (n_samples - nperseg) // step + 1


def freq_vector(nfft: int, fs: float) -> np.ndarray:
    """A7: ``rfftfreq(nfft, 1/fs)`` (scipy:2115) -- ``k * (1/(n*d))`` with d=1/fs."""
    d = 1 / fs
    val = 1.0 / (nfft * d)
    return np.arange(0, nfft // 2 + 1, dtype=int) * val


def time_vector(n_samples: int, nperseg: int, step: int, fs: float) -> np.ndarray:
    """A7: segment-centre times (scipy:2136-2137); ``nperseg/2`` is a float."""
    return np.arange(nperseg / 2, n_samples - nperseg / 2 + 1, step) / float(fs)


# --------------------------------------------------------------------------
# A0..A7 -- the spectrogram itself
# --------------------------------------------------------------------------
def _detrend(frames, kind):
    """A3: scipy.signal.detrend along the last axis (scipy:2070-2072, 2191).

    scipy promotes non-float input to f64 (``_signaltools.detrend``: dtype char
    not in 'dfDF' -> 'd') and keeps float32 as float32.
    """
    if not kind:
        return frames
    if frames.dtype.char not in "dfDF":
        frames = frames.astype(np.float64)
    if kind in ("constant", "c"):
        return frames - np.mean(frames, axis=-1, keepdims=True)
    if kind in ("linear", "l"):
        n = frames.shape[-1]
        # least-squares line over sample index scaled to [1/n .. 1] like scipy
        a = np.ones((n, 2), frames.dtype)
        a[:, 0] = np.arange(1, n + 1, dtype=frames.dtype) / n
        flat = frames.reshape(-1, n).T
        coef, *_ = np.linalg.lstsq(a, flat, rcond=None)
        return (flat - a @ coef).T.reshape(frames.shape)
    raise ValueError("Trend type must be 'linear' or 'constant'.")


def spectrogram(x, fs=1.0, window=("tukey", 0.25), nperseg=None, noverlap=None,
                nfft=None, detrend="constant", return_onesided=True,
                scaling="density", axis=-1, mode="psd"):
    """A0-A7: restatement of ``scipy.signal.spectrogram`` (scipy:816-1005) for a
    real 1-D or batched ``[..., N]`` signal along the last axis.

    Reference mode is the call at PlotEngine.py:113/232: only ``fs`` and
    ``nperseg`` given => Tukey(0.25), ``noverlap = nperseg // 8``,
    ``nfft = nperseg``, constant detrend, one-sided density PSD.
    Returns ``(f, t, Sxx)`` with ``Sxx[..., freq, time]`` exactly like scipy.
    """
    if mode not in ("psd", "complex", "magnitude", "angle", "phase"):
        raise ValueError(f"unknown value for mode {mode!r}")
    if axis != -1:
        raise ValueError("oracle restates the last-axis case only")
    if not return_onesided:
        raise ValueError("oracle restates the one-sided case only")
    x = np.asarray(x)
    if np.iscomplexobj(x):
        raise ValueError("oracle restates real input only")
    n_samples = x.shape[-1]

    # spectrogram() triages first so that the noverlap default sees the clamped nperseg
    win, nperseg = resolve_window(window, nperseg, n_samples)       # scipy:967
    if noverlap is None:
        noverlap = nperseg // 8                                      # scipy:969-970
    noverlap = int(noverlap)
    outdtype = np.result_type(x, np.complex64)                       # scipy:1981
    realdtype = np.float32 if outdtype == np.complex64 else np.float64
    if x.size == 0:
        return np.empty(x.shape), np.empty(x.shape), np.empty(x.shape)
    if nperseg < 1:
        raise ValueError("nperseg must be a positive integer")
    if nfft is None:
        nfft = nperseg
    elif nfft < nperseg:
        raise ValueError("nfft must be greater than or equal to nperseg.")
    nfft = int(nfft)
    if noverlap >= nperseg:
        raise ValueError("noverlap must be less than nperseg.")
    step = nperseg - noverlap

    win = win.astype(realdtype)                                      # scipy:2083-2084
    if scaling == "density":
        scale = 1.0 / (fs * (win * win).sum())                       # scipy:2086-2087
    elif scaling == "spectrum":
        scale = 1.0 / win.sum() ** 2
    else:
        raise ValueError(f"Unknown scaling: {scaling!r}")
    if mode != "psd":
        scale = np.sqrt(scale)

    f = freq_vector(nfft, fs)
    # A2: strided frames
    frames = np.lib.stride_tricks.sliding_window_view(x, nperseg, axis=-1)[..., ::step, :]
    frames = _detrend(frames, detrend)                               # A3
    frames = win * frames                                            # A4
    spec = np.fft.rfft(frames, n=nfft, axis=-1)                      # A5
    if mode == "psd":
        res = np.conjugate(spec) * spec                              # A6
    else:
        res = spec
    res = res * scale
    if mode == "psd":
        if nfft % 2:
            res[..., 1:] *= 2
        else:
            res[..., 1:-1] *= 2                                      # Nyquist not doubled
    t = time_vector(n_samples, nperseg, step, fs)
    res = res.astype(outdtype)
    if mode == "psd":
        res = res.real
    elif mode == "magnitude":
        res = np.abs(res)
    elif mode in ("angle", "phase"):
        res = np.angle(res)
        if mode == "phase":
            # scipy:990-992 unwraps along the FREQUENCY axis of its [..., freq, time] result; here the data is still
            # frame-major [..., time, freq], so that axis is the last one
            res = np.unwrap(res, axis=-1)
    res = np.moveaxis(res, -1, -2)                                   # [..., freq, time]
    return f, t, res


# --------------------------------------------------------------------------
# A8..A10 -- PlotEngine._plot_spectrogram numerics (PlotEngine.py:110-131)
# --------------------------------------------------------------------------
def band_mask(f, fmin, fmax):
    """A8: inclusive-both-ends frequency mask (PlotEngine.py:114, :238)."""
    return (f >= fmin) & (f <= fmax)


def plot_image(f, t, sxx, fmin, fmax, log_scale, global_max=None):
    """A8-A10: what ``_plot_spectrogram`` stores and what it hands to pcolormesh.

    Returns ``(last_f, last_t, last_Sxx, image)``; ``image`` is ``None`` on the
    empty-mask path (PlotEngine.py:122-124, where ``last_t`` becomes ``[]``).
    """
    mask = band_mask(f, fmin, fmax)
    last_f, last_sxx = f[mask].copy(), sxx[mask, :].copy()
    last_t = t.copy()
    if last_sxx.size == 0:
        return last_f, np.array([]), last_sxx, None
    base = np.max(last_sxx) if global_max is None or global_max <= 0 else global_max
    img = np.clip(last_sxx / (base + 1e-20), 0.0, 1.0)               # PlotEngine.py:126-127
    if log_scale:
        db = 10.0 * np.log10(img + 1e-12)                            # PlotEngine.py:129
        db = np.nan_to_num(db)
        lo, hi = np.min(db), np.max(db)
        img = (db - lo) / (hi - lo) if (hi - lo) > 1e-6 else np.zeros_like(db)
    return last_f, last_t, last_sxx, img


# --------------------------------------------------------------------------
# A11 -- PlotEngine._calculate_features (PlotEngine.py:229-242)
# --------------------------------------------------------------------------
def hmm_features(x, fs, nperseg, fmin, fmax):
    """A11: band log-power and its first difference, ``(t, [n_frames, 2])``."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f, t, sxx = spectrogram(x, fs=fs, nperseg=nperseg, scaling="density", mode="psd")
    if sxx.size == 0:
        return None, None
    m = band_mask(f, fmin, fmax)
    power = np.sum(sxx[m, :], axis=0)
    lp = np.log10(power + 1e-20)
    dlp = np.diff(lp, prepend=lp[0])
    return t, np.column_stack([lp, dlp])


# --------------------------------------------------------------------------
# A12 / A13 -- power summaries (PlotEngine.py:686-719)
# --------------------------------------------------------------------------
EEG_BANDS = {
    "Delta (δ)": (0, 4),
    "Theta (θ)": (4, 8),
    "Alpha (α)": (8, 13),
    "Beta (β)": (13, 30),
    "Gamma (γ)": (30, 80),
    "HFO (ripples)": (80, 250),
}


def absolute_power(last_sxx):
    """A12: ``np.sum(last_Sxx)`` or None (PlotEngine.py:686-690)."""
    if last_sxx is None:
        return None
    return np.sum(last_sxx)


def band_powers(last_f, last_sxx, bands=None):
    """A13: relative band powers, half-open ``[lo, hi)`` masks (PlotEngine.py:692-719)."""
    if last_sxx is None or last_f is None:
        return None
    lin = np.maximum(0, last_sxx)
    bands = EEG_BANDS if bands is None else bands
    total = np.sum(lin)
    if total < 1e-18:
        return {name: 0.0 for name in bands}
    out = {}
    for name, (lo, hi) in bands.items():
        m = (last_f >= lo) & (last_f < hi)
        out[name] = np.clip(np.sum(lin[m, :]) / total, 0.0, None)
    return out


# --------------------------------------------------------------------------
# A15 -- SweepManager.get_signal (SweepManager.py:151-185)
# --------------------------------------------------------------------------
def get_signal(data: dict, name: str, processed: bool = False):
    """A15: raw/processed selection with the reference's fs fallbacks and KeyErrors."""
    if name not in data:
        raise KeyError(f"{name} not found in SweepManager.data")
    e = data[name]
    if processed:
        sig = e.get("processed")
        if sig is None:
            sig = e.get("raw")
            if sig is None:
                raise KeyError(f"No 'processed' or 'raw' signal for {name}")
            fs = e.get("fs_raw", e.get("fs"))
        else:
            fs = e.get("fs")
        if fs is None:
            raise KeyError(f"No sampling rate for processed signal of {name}")
        return sig, fs
    sig = e.get("raw")
    if sig is None:
        raise KeyError(f"No 'raw' signal for {name}")
    fs = e.get("fs_raw", e.get("fs"))
    if fs is None:
        raise KeyError(f"No sampling rate for raw signal of {name}")
    return sig, fs
