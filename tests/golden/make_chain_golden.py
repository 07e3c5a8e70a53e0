"""Regenerates tests/golden/spectrum_chain_c1.npz: BASELINE config 1/2 shaped inputs (4096-pt CW
tone + AWGN) pushed through the CPU oracle, whose FFT stage is additionally checked
against the reference's own pocketfft (oracle/_ref, built from /root/reference) before saving.
Run from the repo root:  python tests/golden/make_chain_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle  # noqa: E402
from test_gpu_chain import tone_batch  # noqa: E402

n, b, h, cycles = 4096, 4, 256, 3
x = tone_batch(oracle, b, n, 1234)
stages = oracle.spectrum_chain(x, -100.0, 0.0)
if oracle.have_ref():
    ref = oracle.ref_fft_c2c(stages["product"], axis=1, forward=True)
    assert np.array_equal(ref.view(np.uint32), stages["fft"].view(np.uint32)), "oracle != reference pocketfft"
    print("FFT stage verified bit-exact against the reference's pocketfft")
bins = np.zeros(n * h, np.float32)
for _ in range(cycles):
    oracle.spectrogram(bins, stages["range"], h)
out = os.path.join(os.path.dirname(__file__), "spectrum_chain_c1.npz")
np.savez_compressed(out, x=x, amplitude=stages["amplitude"], range=stages["range"], bins=bins,
                    height=h, cycles=cycles)
print("wrote", out, os.path.getsize(out), "bytes")
