#!/usr/bin/env python3
"""Replays the slot addressing of cyberether_amd/csrc/kernels/fft_quad.hh (the in-place 4096-point exchange) on the CPU:
input element n sits at slot phys(n) = n0 + 8 n1 + 72 n2 + 570 n3 (n = (n3 n2 n1 n0) in base 8); every radix-8 pass reads
eight slots and writes its results back into the same slots; the last pass hands butterfly k's outputs to positions
k + 512 c.  Checked against numpy.fft, and the LDS-DMA piece map (slot pair -> source element pair) against phys().
Test infrastructure: run by tests/test_quad_index_model.py; the product never imports it."""
import numpy as np

N = 4096
S2, S3 = 72, 570
PIECES = 36


def phys(n: int) -> int:
    return (n & 7) + 8 * ((n >> 3) & 7) + S2 * ((n >> 6) & 7) + S3 * (n >> 9)


def piece_source(q: int, lane: int):
    """Source element of the FIRST of the two slots lane `lane` of piece `q` fills, or None for pad slots
    (fft_quad_body: src[m])."""
    e = 128 * q + 2 * lane
    n3, r = divmod(e, S3)
    n2, rr = divmod(r, S2)
    if rr < 64 and n3 < 8:
        return rr + 64 * n2 + 512 * n3
    return None


def transform_in_place(x: np.ndarray) -> np.ndarray:
    w8 = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8)

    def W(k):
        return np.exp(-2j * np.pi * k / N)

    lds = np.zeros(PIECES * 128, complex)
    for q in range(PIECES):  # the LDS-DMA pieces
        for lane in range(64):
            n = piece_source(q, lane)
            if n is not None:
                lds[128 * q + 2 * lane] = x[n]
                lds[128 * q + 2 * lane + 1] = x[n + 1]
    b = np.arange(8)
    for i in range(512):  # pass 0: ido 512, l1 1
        a = phys(i) + S3 * b
        y = w8 @ lds[a]
        lds[a] = y * W(b * i)
    for i in range(64):  # pass 1: ido 64, l1 8
        for k in range(8):
            a = i + S2 * b + S3 * k
            y = w8 @ lds[a]
            lds[a] = y * W(b * 8 * i)
    for i in range(8):  # pass 2: ido 8, l1 64, k = k_lo + 8 k_hi
        for klo in range(8):
            for khi in range(8):
                a = i + 8 * b + S2 * khi + S3 * klo
                y = w8 @ lds[a]
                lds[a] = y * W(b * 64 * i)
    out = np.zeros(N, complex)
    for k in range(512):  # pass 3: ido 1, l1 512
        k0, k1, k2 = k & 7, (k >> 3) & 7, k >> 6
        a = b + 8 * k2 + S2 * k1 + S3 * k0
        out[k + 512 * b] = w8 @ lds[a]
    return out


def bank_conflicts():
    """Worst-case ways per lane group of the kernel's LDS patterns (ds_read_b64: 32-lane groups on slot mod 32;
    ds_write_b64: 16-lane groups on slot mod 16)."""
    def ways(slots, group, mod):
        worst = 1
        for g in range(0, 64, group):
            banks = {}
            for s in slots[g:g + group]:
                banks.setdefault(s % mod, set()).add(s)
            worst = max(worst, max(len(v) for v in banks.values()))
        return worst
    lanes = np.arange(64)
    res = {}
    res["pass0"] = (ways([int(l) for l in lanes], 32, 32), ways([int(l) for l in lanes], 16, 16))
    res["pass1"] = res["pass0"]
    p2 = [int((l & 7) + S2 * (l >> 3)) for l in lanes]
    res["pass2"] = (ways(p2, 32, 32), ways(p2, 16, 16))
    p3 = [int(S3 * (l & 7) + S2 * (l >> 3)) for l in lanes]
    res["pass3_read"] = (ways(p3, 32, 32), None)
    return res


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    print("max |in-place - numpy.fft|:", np.abs(transform_in_place(x) - np.fft.fft(x)).max())
    print("bank conflict ways (read, write):", bank_conflicts())
