#!/usr/bin/env python3
"""Generates tests/golden/pocketfft_ref_vectors.npz from the REFERENCE's own pocketfft
(oracle/_ref/libref_pocketfft.so, compiled in place from /root/reference): forward and backward
c2c outputs for seeded inputs at lengths that exercise every plan kind of pocketfft_c (radix
2/3/4/5/7/8/11 passes, the generic odd radix, Bluestein).  Inputs are regenerated from the seed by
the tests, so only outputs are stored.  Run in the build container (needs /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

LENGTHS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 21, 22, 23, 25, 26, 27, 29, 31, 32,
           33, 35, 37, 39, 44, 46, 47, 49, 50, 51, 53, 55, 58, 59, 62, 64, 65, 74, 77, 82, 91, 94, 97, 98, 100,
           89, 101, 106, 107, 118, 121, 122, 127, 128, 131, 143, 154, 169, 187, 202, 211, 212, 214, 221, 243, 254, 256, 257, 289, 299,
           321, 323,
           343, 360, 422, 512, 529, 539, 625, 810, 847, 1000, 1001, 1024, 1147, 2048, 2209, 2401]


def signal(n: int) -> np.ndarray:
    rng = np.random.default_rng(20260924 + n)
    return (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)


REAL_LENGTHS = [1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 24, 25, 27, 30, 32, 36, 45, 48, 50, 60, 64, 75,
                81, 100, 120, 125, 128, 150, 243, 256, 360, 500, 625, 1000, 1024, 2000,
                191, 199, 211, 257, 401, 523,   # these six: pocketfft_r picks Bluestein
                7, 11, 13, 14, 21, 22, 26, 35, 49, 77, 91, 98, 121, 143, 169, 182, 343, 1001, 2401, 8050]  # radfg / radbg


def real_signal(n: int) -> np.ndarray:
    rng = np.random.default_rng(19700101 + n)
    return rng.standard_normal((2, n)).astype(np.float32)


def main():
    assert oracle.have_ref(), "oracle/_ref not built"
    out = {"lengths": np.array(LENGTHS, np.int64)}
    for n in LENGTHS:
        x = signal(n)
        out[f"fwd_{n}"] = oracle.ref_fft_c2c(x, 1, True)
        out[f"bwd_{n}"] = oracle.ref_fft_c2c(x, 1, False)
        out[f"blue_{n}"] = np.array(oracle.fft_bluestein_size(n), np.int64)
    out["real_lengths"] = np.array(REAL_LENGTHS, np.int64)
    for n in REAL_LENGTHS:
        x = real_signal(n)
        out[f"r2r_fwd_{n}"] = oracle.ref_fft_r2r(x, 1, True)
        out[f"r2r_bwd_{n}"] = oracle.ref_fft_r2r(x, 1, False)
        out[f"r2c_{n}"] = oracle.ref_fft_r2c(x, 1)
        out[f"rblue_{n}"] = np.array(oracle.rfft_bluestein_size(n), np.int64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pocketfft_ref_vectors.npz"), **out)
    kinds = sum(1 for n in LENGTHS if oracle.fft_bluestein_size(n))
    print(f"{len(LENGTHS)} lengths, {kinds} of them Bluestein")


if __name__ == "__main__":
    main()
