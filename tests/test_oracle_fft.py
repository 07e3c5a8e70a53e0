"""The oracle's FFT restatement (pocketfft cfftp, radix 8/4/2) must be BIT-EXACT against the
reference's own vendored pocketfft.hh compiled in place (oracle/_ref), for every power of two
the product supports and beyond, single and batched (pocketfft vectorises over batches)."""
import numpy as np
import pytest


def _signal(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


@pytest.mark.parametrize("m", range(0, 17))
def test_restatement_bit_exact_vs_reference_pocketfft(oracle, m):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    n = 1 << m
    rng = np.random.default_rng(100 + m)
    batch = 5 if n <= 4096 else 2
    x = _signal(rng, (batch, n))
    for forward in (True, False):
        mine = oracle.fft_c2c(x, forward)
        ref = oracle.ref_fft_c2c(x, axis=1, forward=forward)
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), (n, forward)


@pytest.mark.parametrize("n", [3, 5, 6, 9, 10, 12, 15, 20, 25, 30, 45, 60, 75, 100, 125, 243, 250, 360,
                               625, 1000, 2000, 3125, 6000, 10000, 160000])
def test_radix_3_5_restatement_bit_exact_vs_reference_pocketfft(oracle, n):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(n)
    x = _signal(rng, (3 if n < 10000 else 1, n))
    for forward in (True, False):
        mine = oracle.fft_c2c(x, forward)
        ref = oracle.ref_fft_c2c(x, axis=1, forward=forward)
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), (n, forward)


def test_twiddles_match_reference_usage(oracle):
    # exp(+2 pi j k / n) to float accuracy, and exactly symmetric the way pocketfft builds it
    for n in (8, 64, 4096, 65536):
        tw = oracle.fft_twiddles(n)
        k = np.arange(n)
        assert np.max(np.abs(tw - np.exp(2j * np.pi * k / n))) < 1e-7
        assert tw[0] == 1 and np.array_equal(tw[1:n // 2].real, tw[:n // 2:-1].real)


def test_factor_order(oracle):
    assert oracle.fft_factors(4096) == [8, 8, 8, 8]
    assert oracle.fft_factors(8192) == [2, 8, 8, 8, 8]
    assert oracle.fft_factors(2048) == [8, 8, 8, 4]
    assert oracle.fft_factors(65536) == [2, 8, 8, 8, 8, 8]
    assert oracle.fft_factors(12) == [4, 3]
    assert oracle.fft_factors(160000) == [8, 8, 4, 5, 5, 5, 5]
    assert oracle.fft_factors(10) == [2, 5]
    assert oracle.fft_factors(14) == [2, 7]
    assert oracle.fft_factors(8050) == [2, 5, 5, 7, 23]
    assert oracle.fft_bluestein_size(8050) == 0 and oracle.fft_bluestein_size(8292) == 16632
    assert oracle.fft_bluestein_size(97) == 0 and oracle.fft_bluestein_size(101) == 210


@pytest.mark.parametrize("lo", [1, 100, 200, 300, 400])
def test_every_length_bit_exact_vs_reference_pocketfft(oracle, lo):
    """All lengths 1..499 -- radix 7 / 11 passes, the generic odd radix and Bluestein included."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(lo)
    for n in range(lo, lo + 100):
        x = _signal(rng, (2, n))
        for forward in (True, False):
            mine = oracle.fft_c2c(x, forward)
            ref = oracle.ref_fft_c2c(x, axis=1, forward=forward)
            assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), (n, forward)


@pytest.mark.parametrize("n", [4099, 8050, 8191, 8292, 10007, 12345, 30030, 65521])
def test_large_odd_plans_bit_exact_vs_reference_pocketfft(oracle, n):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    x = _signal(np.random.default_rng(n), (1, n))
    for forward in (True, False):
        assert np.array_equal(oracle.fft_c2c(x, forward).view(np.uint32),
                              oracle.ref_fft_c2c(x, axis=1, forward=forward).view(np.uint32)), (n, forward)


def test_restatement_matches_committed_reference_vectors(oracle):
    """Travels to the GPU box: outputs of the reference's own pocketfft, generated here by
    tests/golden/make_pocketfft_golden.py, for every plan kind."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_pocketfft_golden import signal
    gold = np.load(os.path.join(here, "golden", "pocketfft_ref_vectors.npz"))
    blue = 0
    for n in gold["lengths"]:
        n = int(n)
        x = signal(n)
        assert oracle.fft_bluestein_size(n) == int(gold[f"blue_{n}"]), n
        blue += int(gold[f"blue_{n}"]) != 0
        for key, fwd in (("fwd", True), ("bwd", False)):
            assert np.array_equal(oracle.fft_c2c(x, fwd).view(np.uint32), gold[f"{key}_{n}"].view(np.uint32)), (n, key)
    assert blue >= 8


def test_real_transforms_bit_exact_vs_reference_pocketfft(oracle):
    """rfftp (radices 2/3/4/5 and the generic radfg/radbg) and the real Bluestein path vs the
    reference: every length up to 420 plus large ones."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(11)
    supported = 0
    for n in list(range(1, 420)) + [1000, 1024, 4096, 6000, 8100, 16000, 65536, 4099, 8191, 8050, 10007, 30030]:
        x = rng.standard_normal((2, n)).astype(np.float32)
        fwd = oracle.fft_r2r(x, True)
        supported += 1
        assert np.array_equal(fwd.view(np.uint32), oracle.ref_fft_r2r(x, 1, True).view(np.uint32)), n
        assert np.array_equal(oracle.fft_r2r(x, False).view(np.uint32),
                              oracle.ref_fft_r2r(x, 1, False).view(np.uint32)), n
        assert np.array_equal(oracle.fft_r2c(x).view(np.uint32), oracle.ref_fft_r2c(x, 1).view(np.uint32)), n
    assert supported >= 430


def test_real_restatement_matches_committed_reference_vectors(oracle):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_pocketfft_golden import real_signal
    gold = np.load(os.path.join(here, "golden", "pocketfft_ref_vectors.npz"))
    blue = 0
    for n in gold["real_lengths"]:
        n = int(n)
        x = real_signal(n)
        assert oracle.rfft_bluestein_size(n) == int(gold[f"rblue_{n}"]), n
        blue += int(gold[f"rblue_{n}"]) != 0
        assert np.array_equal(oracle.fft_r2r(x, True).view(np.uint32), gold[f"r2r_fwd_{n}"].view(np.uint32)), n
        assert np.array_equal(oracle.fft_r2r(x, False).view(np.uint32), gold[f"r2r_bwd_{n}"].view(np.uint32)), n
        assert np.array_equal(oracle.fft_r2c(x).view(np.uint32), gold[f"r2c_{n}"].view(np.uint32)), n
    assert blue >= 5
