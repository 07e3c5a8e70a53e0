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
    assert oracle.fft_factors(14) is None  # pass7 is not restated
