"""CPU restatement of the mel stage -- TEST INFRASTRUCTURE ONLY.

The reference has NO mel filterbank (grep: zero hits), so there is nothing to pin this against:
**parity unpinned**.  The definition is this build's own (documented in include/spectro.h, sg_mel_weights):
HTK mel scale, n_mels triangular filters with unit peak (no area normalisation) between fmin and fmax,
evaluated at the rfft bin frequencies.  The GPU stage is verified against this restatement only.
"""
import numpy as np


def hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, np.float64) / 700.0)


def mel_to_hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, np.float64) / 2595.0) - 1.0)


def mel_weights(nfft, fs, n_mels, fmin, fmax):
    """Dense ``[nfft//2+1, n_mels]`` float64 filterbank."""
    f = np.arange(nfft // 2 + 1) * fs / nfft
    edges = mel_to_hz(hz_to_mel(fmin) + (hz_to_mel(fmax) - hz_to_mel(fmin)) * np.arange(n_mels + 2) / (n_mels + 1))
    l, c, r = edges[:-2], edges[1:-1], edges[2:]
    up = (f[:, None] - l[None, :]) / (c - l)[None, :]
    down = (r[None, :] - f[:, None]) / (r - c)[None, :]
    return np.maximum(0.0, np.minimum(up, down))


def mel_spectrogram(sxx_frame_major, weights, log_scale=False):
    """``[..., frames, bins] @ [bins, mels]`` in float64, optional 10*log10(max(x, 1e-10))."""
    m = np.asarray(sxx_frame_major, np.float64) @ np.asarray(weights, np.float64)
    return 10.0 * np.log10(np.maximum(m, 1e-10)) if log_scale else m
