#!/usr/bin/env python3
"""Lane-level numpy model of the smallest-transform register kernel (stft_rtiny.hip: nfft = 64 / 32, Q = 4 / 2 lanes per frame).

Lane Q g + j holds z_g[j + Q r], r < 8: radix-8 in registers, per-lane twiddles, a Q-point DFT across the quad's lanes by DPP quad
permutes, the real-input split against the mirror lane.  Checks the index maps and the stage-B coefficients (CPU)."""
import numpy as np

w8 = np.exp(-2j * np.pi * np.arange(8)[:, None] * np.arange(8)[None, :] / 8)


def quad(v, perm):
    lane = np.arange(64)
    return v[(lane & ~3) + np.asarray(perm)[lane & 3]]


def run_general(Q):
    """Q no power of two (nperseg 96 / 160 / 192 / 224): the cross-lane DFT as a direct sum, the split's partner in the mirror lane of the frame"""
    G, N2 = 64 // Q, 8 * Q
    n = 2 * N2
    rng = np.random.default_rng(Q)
    x = rng.standard_normal((G, n))
    z = x[:, 0::2] + 1j * x[:, 1::2]
    lane = np.arange(G * Q)
    j, g = lane % Q, lane // Q
    a = np.stack([z[g, j + Q * r] for r in range(8)], axis=1) @ w8
    a = a * np.exp(-2j * np.pi * j[:, None] * np.arange(8)[None, :] / N2)
    out = np.zeros_like(a)
    for k1 in range(8):
        for l in lane:
            out[l, k1] = sum(a[g[l] * Q + jj, k1] * np.exp(-2j * np.pi * ((jj * j[l]) % Q) / Q) for jj in range(Q))
    a, k2 = out, j
    assert np.allclose(a, np.fft.fft(z, axis=1)[g[:, None], np.arange(8)[None, :] + 8 * k2[:, None]])
    mirror, mirror0 = g * Q + (Q - 1 - j), g * Q + (Q - j) % Q
    P = np.zeros((G, N2 + 1))
    for k1 in range(8):
        A = a[:, k1]
        B = a[mirror0, 0] if k1 == 0 else a[mirror, 8 - k1]
        k = k1 + 8 * k2
        tw = np.exp(-2j * np.pi * k / n)
        S = (A.real + B.real) + 1j * (A.imag - B.imag)
        D = (A.real - B.real) + 1j * (A.imag + B.imag)
        X = (S.real + tw.real * D.imag + tw.imag * D.real) + 1j * (S.imag + tw.imag * D.imag - tw.real * D.real)
        P[g, k] = np.abs(X) ** 2 / 4
    sel = k2 == 0
    P[g[sel], N2] = (a[sel, 0].real - a[sel, 0].imag) ** 2
    ref = np.abs(np.fft.rfft(x, axis=1)) ** 2
    return np.abs(P - ref).max() / ref.max()


def run(Q):
    G, N2 = 64 // Q, 8 * Q
    n = 2 * N2
    rng = np.random.default_rng(Q)
    x = rng.standard_normal((G, n))
    z = x[:, 0::2] + 1j * x[:, 1::2]
    lane = np.arange(64)
    j, g = lane % Q, lane // Q
    k2 = np.where(Q == 4, ((j & 1) << 1) | (j >> 1), j)
    a = np.stack([z[g, j + Q * r] for r in range(8)], axis=1)              # [lane, r]
    a = a @ w8                                                             # -> k1
    a = a * np.exp(-2j * np.pi * j[:, None] * np.arange(8)[None, :] / N2)
    sA = np.where((j & 2) != 0, -1.0, 1.0) if Q == 4 else np.where(j != 0, -1.0, 1.0)
    cA = np.array([1, -1, 1, 1j])[j] if Q == 4 else None
    cB = np.array([1, 1, -1j, 1])[j] if Q == 4 else None
    X1, X2, MIR, SWH = [1, 0, 3, 2], [2, 3, 0, 1], [3, 2, 1, 0], [0, 1, 3, 2]
    for k1 in range(8):
        pa = quad(a[:, k1], X2 if Q == 4 else X1)
        u = pa + sA * a[:, k1]
        a[:, k1] = u * cA + quad(u, X1) * cB if Q == 4 else u
    ref_z = np.fft.fft(z, axis=1)
    for k1 in range(8):
        assert np.allclose(a[:, k1], ref_z[g, k1 + 8 * k2]), ("Z", Q, k1)
    P = np.zeros((G, N2 + 1))
    for k1 in range(8):
        A = a[:, k1]
        if k1 == 0:
            B = quad(a[:, 0], SWH) if Q == 4 else a[:, 0]
        else:
            B = quad(a[:, 8 - k1], MIR if Q == 4 else X1)
        k = k1 + 8 * k2
        tw = np.exp(-2j * np.pi * k / n)
        S = (A.real + B.real) + 1j * (A.imag - B.imag)
        D = (A.real - B.real) + 1j * (A.imag + B.imag)
        X = (S.real + tw.real * D.imag + tw.imag * D.real) + 1j * (S.imag + tw.imag * D.imag - tw.real * D.real)
        P[g, k] = np.abs(X) ** 2 / 4
    sel = k2 == 0
    P[g[sel], N2] = (a[sel, 0].real - a[sel, 0].imag) ** 2
    ref = np.abs(np.fft.rfft(x, axis=1)) ** 2
    return np.abs(P - ref).max() / ref.max()


if __name__ == "__main__":
    for Q in (4, 2):
        print("Q", Q, "nfft", 16 * Q, "max rel err", run(Q))
    for Q in (6, 10, 12, 14):
        print("Q", Q, "nfft", 16 * Q, "max rel err", run_general(Q))
