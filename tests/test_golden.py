"""CPU: the committed golden fixture is what the oracle produces today (guards against silent
drift of either); the GPU side is tests/test_gpu_chain.py::test_golden_fixture."""
import os

import numpy as np


def test_oracle_reproduces_golden(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "spectrum_chain_c1.npz"))
    stages = oracle.spectrum_chain(g["x"], -100.0, 0.0)
    assert np.array_equal(stages["amplitude"].view(np.uint32), g["amplitude"].view(np.uint32))
    assert np.array_equal(stages["range"].view(np.uint32), g["range"].view(np.uint32))
    bins = np.zeros_like(g["bins"])
    for _ in range(int(g["cycles"])):
        oracle.spectrogram(bins, stages["range"], int(g["height"]))
    assert np.array_equal(bins.view(np.uint32), g["bins"].view(np.uint32))
    # sanity of the physics: the tone of row r peaks at the centred bin n/2 + 100 + r
    n = g["x"].shape[1]
    assert [int(np.argmax(r)) for r in stages["range"]] == [n // 2 + 100 + i for i in range(4)]
