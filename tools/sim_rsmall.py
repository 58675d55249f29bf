#!/usr/bin/env python3
"""Lane-level numpy model of the small-transform register kernel (nfft = 128*R, R in {2,4}: 256 / 512).

A wave carries G = 8/R frames at once: lane j holds z_g[j + 64a] (a < R) for every frame g, i.e. 8 complex values
per lane like the 1024 kernel.  Pass 1 is an R-point DFT per frame, passes 2/3 are the radix-8 passes of
stft_r8x3 with the register index v = g*R + r.  Design aid: checks the index maps and LDS bank behaviour."""
import numpy as np

S1, S2 = 72, 66
w8 = np.exp(-2j * np.pi * np.arange(8)[:, None] * np.arange(8)[None, :] / 8)


def banks_write_b64(a):
    return max(np.bincount(a[16 * g:16 * g + 16] % 16, minlength=16).max() for g in range(4))


def banks_read_b64(a):
    worst = 1
    for g in range(2):
        x = a[32 * g:32 * g + 32]
        worst = max(worst, max(len(set(x[x % 32 == b])) for b in range(32)))
    return worst


def run(R, rep):
    G, M = 8 // R, 64 * R
    L = 8 * R                       # lanes per frame in pass 3
    rng = np.random.default_rng(R)
    x = rng.standard_normal((G, 2 * M))
    z = x[:, 0::2] + 1j * x[:, 1::2]
    lane = np.arange(64)
    wR = np.exp(-2j * np.pi * np.arange(R)[:, None] * np.arange(R)[None, :] / R)
    reg = np.zeros((64, 8), complex)
    for g in range(G):
        v = np.stack([z[g, lane + 64 * a] for a in range(R)], axis=1) @ wR               # [lane, r]
        v = v * np.exp(-2j * np.pi * lane[:, None] * np.arange(R)[None, :] / M)
        reg[:, g * R:(g + 1) * R] = v
    # exchange 1 (identical to r8x3 with v as the register index)
    lds = np.zeros(8 * S1, complex)
    j0, b = lane % 8, lane // 8
    for v in range(8):
        addr = b * S1 + j0 + 8 * v
        rep["x1w"] = max(rep.get("x1w", 1), banks_write_b64(addr)); lds[addr] = reg[:, v]
    new = np.zeros((64, 8), complex)
    for bb in range(8):
        addr = bb * S1 + lane
        rep["x1r"] = max(rep.get("x1r", 1), banks_read_b64(addr)); new[:, bb] = lds[addr]
    reg = (new @ w8) * np.exp(-2j * np.pi * (lane % 8)[:, None] * np.arange(8)[None, :] / 64)
    # exchange 2: lane l2 = j0 + 8v, v = g*R + r ; dest lane l3 = (r + R*s) + L*g, slot j0
    lds = np.zeros(8 * S2, complex)
    v_ = lane // 8
    g_, r_ = v_ // R, v_ % R
    for s in range(8):
        l3 = r_ + R * s + L * g_
        addr = j0 * S2 + l3
        rep["x2w"] = max(rep.get("x2w", 1), banks_write_b64(addr)); lds[addr] = reg[:, s]
    new = np.zeros((64, 8), complex)
    for jj in range(8):
        addr = jj * S2 + lane
        rep["x2r"] = max(rep.get("x2r", 1), banks_read_b64(addr)); new[:, jj] = lds[addr]
    reg = new @ w8                                           # lane l3 = lu + L*g holds Z_g[lu + L*t]
    # split: upper half through LDS, per frame region of M+1 (+pad to keep regions apart)
    RS = M + 8
    Zl = np.zeros(G * RS, complex)
    g3, lu = lane // L, lane % L
    for t in range(4, 8):
        addr = g3 * RS + lu + L * t
        rep["x3w"] = max(rep.get("x3w", 1), banks_write_b64(addr)); Zl[addr] = reg[:, t]
    for g in range(G):
        Zl[g * RS + M] = reg[L * g, 0]                        # Z[M] := Z[0]
    P = np.zeros((G, M + 1))
    for t in range(4):
        k = lu + L * t
        addr = g3 * RS + M - k
        rep["x3r"] = max(rep.get("x3r", 1), banks_read_b64(addr))
        A, B = reg[:, t], np.conj(Zl[addr])
        T = 1j * np.exp(-2j * np.pi * k / (2 * M)) * (A - B)
        P[g3, k] = np.abs((A + B) - T) ** 2 / 4
        P[g3, M - k] = np.abs((A + B) + T) ** 2 / 4
    for g in range(G):
        P[g, M // 2] = np.abs(reg[L * g, 4]) ** 2              # k = M/2: lane lu = 0, register t = 4
    ref = np.abs(np.fft.rfft(x, axis=1)) ** 2
    return np.abs(P - ref).max() / ref.max()


if __name__ == "__main__":
    for R in (1, 2, 4, 8):
        rep = {}
        print("R", R, "nfft", 128 * R, "max rel err", run(R, rep), rep)
