#!/usr/bin/env python3
"""Lane-level numpy model of the large-transform register kernel (nfft = 128*R, R = 8T, T in {2,4}: 2048 / 4096).

Lane j holds z[j + 64a], a < R.  Pass 1 = in-register R-point DFT (radix-8 over a1, constant twiddle, radix-T over a0),
passes 2/3 = T radix-8 butterflies per lane.  Final layout: lane l holds Z[l + 64c], c < R.  Design aid."""
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


def run(T, rep):
    R, M = 8 * T, 64 * 8 * T
    rng = np.random.default_rng(T)
    x = rng.standard_normal(2 * M)
    z = x[0::2] + 1j * x[1::2]
    lane = np.arange(64)
    wT = np.exp(-2j * np.pi * np.arange(T)[:, None] * np.arange(T)[None, :] / T)
    reg = np.stack([z[lane + 64 * a] for a in range(R)], axis=1)                  # [lane, a]
    # pass 1: a = a0 + T*a1
    u = np.zeros((64, T, 8), complex)
    for a0 in range(T):
        u[:, a0, :] = reg[:, a0::T] @ w8                                             # over a1 -> r1
        u[:, a0, :] *= np.exp(-2j * np.pi * a0 * np.arange(8) / R)[None, :]
    Y = np.zeros((64, R), complex)
    for r1 in range(8):
        Y[:, r1::8] = u[:, :, r1] @ wT                                               # over a0 -> r0 ; r = r1 + 8*r0
    Y *= np.exp(-2j * np.pi * lane[:, None] * np.arange(R)[None, :] / M)
    # exchange 1
    lds = np.zeros(T * 8 * S1, complex)
    j0, b = lane % 8, lane // 8
    for r in range(R):
        r1, q = r % 8, r // 8
        addr = q * 8 * S1 + b * S1 + j0 + 8 * r1
        rep["x1w"] = max(rep.get("x1w", 1), banks_write_b64(addr)); lds[addr] = Y[:, r]
    reg2 = np.zeros((64, T, 8), complex)
    for q in range(T):
        for bb in range(8):
            addr = q * 8 * S1 + bb * S1 + lane
            rep["x1r"] = max(rep.get("x1r", 1), banks_read_b64(addr)); reg2[:, q, bb] = lds[addr]
    # pass 2: lane l2 = j0 + 8*r1, tasks q
    for q in range(T):
        reg2[:, q, :] = (reg2[:, q, :] @ w8) * np.exp(-2j * np.pi * (lane % 8)[:, None] * np.arange(8)[None, :] / 64)
    # exchange 2: value (q, s) of lane (j0, r1): r = r1 + 8q ; u = r + R*s ; l3 = u % 64 ; q3 = u // 64
    lds = np.zeros(T * 8 * S2, complex)
    r1_ = lane // 8
    for q in range(T):
        for s in range(8):
            uu = r1_ + 8 * q + R * s
            addr = (uu // 64) * 8 * S2 + j0 * S2 + (uu % 64)
            rep["x2w"] = max(rep.get("x2w", 1), banks_write_b64(addr)); lds[addr] = reg2[:, q, s]
    reg3 = np.zeros((64, T, 8), complex)
    for q3 in range(T):
        for jj in range(8):
            addr = q3 * 8 * S2 + jj * S2 + lane
            rep["x2r"] = max(rep.get("x2r", 1), banks_read_b64(addr)); reg3[:, q3, jj] = lds[addr]
    Z = np.zeros((64, R), complex)                                                    # Z[lane, c] = Z[l + 64c], c = q3 + T*t
    for q3 in range(T):
        Z[:, q3::T] = reg3[:, q3, :] @ w8
    # check the FFT itself
    ref = np.fft.fft(z)
    got = np.zeros(M, complex)
    for c in range(R):
        got[lane + 64 * c] = Z[:, c]
    err_fft = np.abs(got - ref).max() / np.abs(ref).max()
    # split: upper half through LDS
    Zl = np.zeros(M + 1, complex)
    for c in range(R // 2, R):
        addr = lane + 64 * c
        rep["x3w"] = max(rep.get("x3w", 1), banks_write_b64(addr)); Zl[addr] = Z[:, c]
    Zl[M] = Z[0, 0]
    P = np.zeros(M + 1)
    for c in range(R // 2):
        k = lane + 64 * c
        rep["x3r"] = max(rep.get("x3r", 1), banks_read_b64(M - k))
        A, B = Z[:, c], np.conj(Zl[M - k])
        Tt = 1j * np.exp(-2j * np.pi * k / (2 * M)) * (A - B)
        P[k] = np.abs((A + B) - Tt) ** 2 / 4
        P[M - k] = np.abs((A + B) + Tt) ** 2 / 4
    P[M // 2] = np.abs(Z[0, R // 2]) ** 2
    refp = np.abs(np.fft.rfft(x)) ** 2
    return err_fft, np.abs(P - refp).max() / refp.max()


if __name__ == "__main__":
    for T in (1, 2, 4):
        rep = {}
        print("T", T, "nfft", 1024 * T, "err", run(T, rep), rep)
