#!/usr/bin/env python3
"""Lane-level numpy model of the n_fft=1024 register kernel (64 lanes x 8 complex, radix-8 x3).

Design aid only (not imported by the product or the tests): it replays the exact
register/LDS index maps of csrc/stft_r8.hip so that the maps and the LDS bank
behaviour can be checked on the CPU before touching a GPU.
"""
import numpy as np

M = 512
S1, S2 = 72, 66
w8 = np.exp(-2j * np.pi * np.arange(8)[:, None] * np.arange(8)[None, :] / 8)


def banks_write_b64(addr_elems):
    """ds_write_b64: 4 groups of 16 contiguous lanes, 32 banks of 4 B -> 16 element-banks."""
    worst = 1
    for g in range(4):
        e = addr_elems[16 * g:16 * g + 16] % 16
        worst = max(worst, np.bincount(e, minlength=16).max())
    return worst


def banks_read_b64(addr_elems):
    """ds_read_b64: 2 groups of 32 lanes, 64 banks -> 32 element-banks."""
    worst = 1
    for g in range(2):
        a = addr_elems[32 * g:32 * g + 32]
        e = a % 32
        # identical addresses broadcast
        worst = max(worst, max(len(set(a[e == b])) for b in range(32)))
    return worst


def fft512_lanes(z, report):
    lane = np.arange(64)
    # pass 1: lane j holds z[j + 64a]
    reg = np.stack([z[lane + 64 * a] for a in range(8)], axis=1)          # [lane, a]
    reg = reg @ w8                                                         # [lane, r]
    reg = reg * np.exp(-2j * np.pi * lane[:, None] * np.arange(8)[None, :] / 512)
    # exchange 1: lane (j0,b) -> LDS[b*S1 + j0 + 8r]; lane l2 reads [b*S1 + l2]
    lds = np.zeros(8 * S1, complex)
    j0, b = lane % 8, lane // 8
    for r in range(8):
        addr = b * S1 + j0 + 8 * r
        report["x1 write"] = max(report.get("x1 write", 1), banks_write_b64(addr))
        lds[addr] = reg[:, r]
    new = np.zeros((64, 8), complex)
    for bb in range(8):
        addr = bb * S1 + lane
        report["x1 read"] = max(report.get("x1 read", 1), banks_read_b64(addr))
        new[:, bb] = lds[addr]
    reg = new                                                              # lane l2 = j0 + 8r, slot b
    # pass 2
    reg = reg @ w8                                                         # [lane, s]
    j0 = lane % 8
    reg = reg * np.exp(-2j * np.pi * j0[:, None] * np.arange(8)[None, :] / 64)
    # exchange 2: lane (j0, r) -> LDS[j0*S2 + r + 8s]; lane l3 reads [j0*S2 + l3]
    lds = np.zeros(8 * S2, complex)
    r_ = lane // 8
    for s in range(8):
        addr = j0 * S2 + r_ + 8 * s
        report["x2 write"] = max(report.get("x2 write", 1), banks_write_b64(addr))
        lds[addr] = reg[:, s]
    new = np.zeros((64, 8), complex)
    for jj in range(8):
        addr = jj * S2 + lane
        report["x2 read"] = max(report.get("x2 read", 1), banks_read_b64(addr))
        new[:, jj] = lds[addr]
    reg = new @ w8                                                         # lane l3 = r + 8s, slot t
    # exchange 3: natural order
    Z = np.zeros(M + 1, complex)
    for t in range(8):
        addr = lane + 64 * t
        report["x3 write"] = max(report.get("x3 write", 1), banks_write_b64(addr))
        Z[addr] = reg[:, t]
    Z[M] = Z[0]
    return Z


def split_power(Z, report):
    lane = np.arange(64)
    P = np.zeros(M + 1)
    for m in range(4):
        k = lane + 64 * m
        report["x3 read"] = max(report.get("x3 read", 1), banks_read_b64(k), banks_read_b64(M - k))
        A, B = Z[k], np.conj(Z[M - k])
        Wk = np.exp(-2j * np.pi * k / 1024)
        T = 1j * Wk * (A - B)
        P[k] = np.abs((A + B) - T) ** 2 / 4
        P[M - k] = np.abs((A + B) + T) ** 2 / 4
    P[256] = np.abs(Z[256]) ** 2
    return P


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1024)
    rep = {}
    Z = fft512_lanes(x[0::2] + 1j * x[1::2], rep)
    P = split_power(Z, rep)
    ref = np.abs(np.fft.rfft(x)) ** 2
    print("max rel err", np.abs(P - ref).max() / ref.max())
    print("bank conflict ways:", rep)
