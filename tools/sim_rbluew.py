#!/usr/bin/env python3
"""numpy model of the wide register kernels' frame algebra (stft_rbluew.hip / stft_rbluew_f64.hip), CPU:
the real frame packed into N2 complex points, decimation in time over W waves (each a chirp-z transform of mp = N2 / W points on two
L-point FFTs -- or, for nperseg 8192, ONE plain mp-point FFT), the radix-W combine, the real-input split with the split twiddle as a per-lane
times a per-row factor, W_N2^k0 raised to the power w."""
import numpy as np


def chirpz(z, L):
    mp = z.size
    a = np.arange(mp)
    c = np.exp(-1j * np.pi * ((a * a) % (2 * mp)) / mp)
    b = np.zeros(L, complex)
    b[:mp] = np.conj(c)
    b[L - mp + 1:] = np.conj(c[1:][::-1])
    B = np.fft.fft(b) / L
    y = np.zeros(L, complex)
    y[:mp] = z * c
    V = np.fft.fft(np.conj(np.fft.fft(y) * B))               # the inverse transform as a forward one between two conjugations
    return c * np.conj(V[:mp])


def frame(x, W, L, exact=False):
    n = x.size
    n2, mp = n // 2, n // 2 // W
    z = x[0::2] + 1j * x[1::2]
    k0 = np.arange(mp)
    ctw = np.exp(-2j * np.pi * k0 / n2)
    G = []
    for w in range(W):
        zw = z[w::W]
        F = np.fft.fft(zw) if exact else chirpz(zw, L)
        G.append(F * ctw ** w)
    Z = np.zeros(n2 + 1, complex)
    for r in range(W):
        Z[r * mp:(r + 1) * mp] = sum(np.exp(-2j * np.pi * v * r / W) * G[v] for v in range(W))
    Z[n2] = Z[0]
    k = np.arange(n2 + 1)
    lane, row = k % 64, k // 64
    tw = np.exp(-2j * np.pi * lane / n) * np.exp(-2j * np.pi * 64 * row / n)
    A, B = Z[k], Z[n2 - k]
    S = (A.real + B.real) + 1j * (A.imag - B.imag)
    D = (A.real - B.real) + 1j * (A.imag + B.imag)
    X = (S.real + tw.real * D.imag + tw.imag * D.real) + 1j * (S.imag + tw.imag * D.imag - tw.real * D.real)
    return np.abs(X) ** 2 / 4


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n, W, L, exact in [(3000, 2, 2048, False), (4064, 2, 2048, False), (6000, 4, 2048, False), (8160, 4, 2048, False), (8192, 2, 2048, True),
                           (2000, 2, 1024, False), (4000, 4, 1024, False), (8000, 8, 1024, False), (8192, 4, 1024, True)]:
        x = rng.standard_normal(n)
        ref = np.abs(np.fft.rfft(x)) ** 2
        print(f"nperseg {n} W {W} L {L}{' exact' if exact else ''}: max rel err {np.abs(frame(x, W, L, exact) - ref).max() / ref.max():.2e}")
