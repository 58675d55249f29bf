"""Window tables for the STFT plan (host side, float64).

Own implementation of the ``scipy.signal.get_window`` subset the engine supports; the
reference only ever requests scipy's default ``('tukey', 0.25)`` (PlotEngine.py:113 passes
no window), the BASELINE configs add ``'hann'``.  ``fftbins=True`` (periodic) like
``_triage_segments`` uses (scipy/signal/_spectral_py.py:2251).
"""
from __future__ import annotations

import numpy as np

__all__ = ["get_window"]


def _cosine_sum(m, coeffs):
    fac = np.linspace(-np.pi, np.pi, m)
    w = np.zeros(m)
    for k, a in enumerate(coeffs):
        w += a * np.cos(k * fac)
    return w


def _tukey(m, alpha=0.5):
    if alpha <= 0:
        return np.ones(m, "d")
    if alpha >= 1.0:
        return _cosine_sum(m, [0.5, 0.5])
    n = np.arange(0, m)
    width = int(np.floor(alpha * (m - 1) / 2.0))
    n1, n2, n3 = n[0:width + 1], n[width + 1:m - width - 1], n[m - width - 1:]
    w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (m - 1))))
    w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (m - 1))))
    return np.concatenate((w1, np.ones(n2.shape), w3))


def _triang(m):
    n = np.arange(1, (m + 1) // 2 + 1)
    if m % 2 == 0:
        w = (2 * n - 1.0) / m
        return np.r_[w, w[::-1]]
    w = 2 * n / (m + 1.0)
    return np.r_[w, w[-2::-1]]


def _bartlett(m):
    n = np.arange(0, m)
    return np.where(np.less_equal(n, (m - 1) / 2.0), 2.0 * n / (m - 1), 2.0 - 2.0 * n / (m - 1))


def _kaiser(m, beta):
    n = np.arange(0, m)
    alpha = (m - 1) / 2.0
    return np.i0(beta * np.sqrt(1 - ((n - alpha) / alpha) ** 2.0)) / np.i0(beta)


def _gaussian(m, std):
    n = np.arange(0, m) - (m - 1.0) / 2.0
    return np.exp(-n ** 2 / (2 * std * std))


_SYMMETRIC = {
    "boxcar": lambda m: np.ones(m), "box": lambda m: np.ones(m), "ones": lambda m: np.ones(m),
    "rect": lambda m: np.ones(m), "rectangular": lambda m: np.ones(m),
    "triang": _triang, "triangle": _triang, "tri": _triang,
    "bartlett": _bartlett, "bart": _bartlett, "brt": _bartlett,
    "hann": lambda m: _cosine_sum(m, [0.5, 0.5]), "hanning": lambda m: _cosine_sum(m, [0.5, 0.5]),
    "han": lambda m: _cosine_sum(m, [0.5, 0.5]),
    "hamming": lambda m: _cosine_sum(m, [0.54, 0.46]), "hamm": lambda m: _cosine_sum(m, [0.54, 0.46]),
    "ham": lambda m: _cosine_sum(m, [0.54, 0.46]),
    "blackman": lambda m: _cosine_sum(m, [0.42, 0.50, 0.08]), "black": lambda m: _cosine_sum(m, [0.42, 0.50, 0.08]),
    "blk": lambda m: _cosine_sum(m, [0.42, 0.50, 0.08]),
    "nuttall": lambda m: _cosine_sum(m, [0.3635819, 0.4891775, 0.1365995, 0.0106411]),
    "nutl": lambda m: _cosine_sum(m, [0.3635819, 0.4891775, 0.1365995, 0.0106411]),
    "nut": lambda m: _cosine_sum(m, [0.3635819, 0.4891775, 0.1365995, 0.0106411]),
    "blackmanharris": lambda m: _cosine_sum(m, [0.35875, 0.48829, 0.14128, 0.01168]),
    "blackharr": lambda m: _cosine_sum(m, [0.35875, 0.48829, 0.14128, 0.01168]),
    "bkh": lambda m: _cosine_sum(m, [0.35875, 0.48829, 0.14128, 0.01168]),
    "flattop": lambda m: _cosine_sum(m, [0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368]),
    "flat": lambda m: _cosine_sum(m, [0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368]),
    "flt": lambda m: _cosine_sum(m, [0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368]),
    "cosine": lambda m: np.sin(np.pi / m * (np.arange(0, m) + .5)),
    "halfcosine": lambda m: np.sin(np.pi / m * (np.arange(0, m) + .5)),
    "tukey": _tukey, "tuk": _tukey,
    "kaiser": _kaiser, "ksr": _kaiser,
    "gaussian": _gaussian, "gauss": _gaussian, "gss": _gaussian,
    "general_hamming": lambda m, a: _cosine_sum(m, [a, 1.0 - a]),
    "general_cosine": lambda m, a: _cosine_sum(m, list(a)),
}
_NEEDS_PARAMS = {"kaiser", "ksr", "gaussian", "gauss", "gss", "general_hamming", "general_cosine"}


def get_window(window, nx: int, fftbins: bool = True) -> np.ndarray:
    """``scipy.signal.get_window`` for the supported names; float beta = Kaiser like scipy."""
    if isinstance(window, tuple):
        name, args = window[0], tuple(window[1:])
    elif isinstance(window, str):
        name, args = window, ()
    else:
        try:
            name, args = "kaiser", (float(window),)
        except (TypeError, ValueError) as e:
            raise ValueError(f"{type(window)} as window type is not supported.") from e
    if name not in _SYMMETRIC:
        raise ValueError("Unknown window type.")
    if name in _NEEDS_PARAMS and not args:
        raise ValueError(f"The '{name}' window needs one or more parameters -- pass a tuple.")
    nx = int(nx)
    if nx < 0:
        raise ValueError("Window length M must be a non-negative integer")
    if nx <= 1:
        return np.ones(nx)
    m = nx + 1 if fftbins else nx
    w = _SYMMETRIC[name](m, *args)
    return w[:-1] if fftbins else w
