"""provider="fast" (hardware sqrt/exp/rcp instead of the libm restatement): float outputs stay
within the tolerance BASELINE.json states for float spectra (1e-5 of the output range; measured
bounds asserted below are far tighter), and the integer bin work downstream stays exact."""
import numpy as np
import pytest

from test_gpu_chain import tone_batch
from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu

AMPLITUDE_TOL_DB = 5e-6   # |dB error| (1-ulp sqrt ahead of the reference's own cubic log10)
RANGE_TOL_ABS = 3e-7      # |error| of the [0,1] range output; BASELINE tolerance is 1e-5


@pytest.mark.parametrize("fuse", [True, False])
def test_fast_chain_within_tolerance_and_bins_exact(js, oracle, fuse):
    n, b, h = 4096, 24, 256
    x = tone_batch(oracle, b, n, 31)
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src, provider="fast")
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer})
    rt = js.Runtime(eng.modules + [spec], fuse=fuse, graph=True)
    assert any(u.startswith("spectrum_fused") for u in rt.units) == fuse
    rt.compute(2)
    ref = oracle.spectrum_chain(x, -100.0, 0.0)
    got = eng.buffer.numpy()
    assert np.max(np.abs(got - ref["range"])) <= RANGE_TOL_ABS
    if not fuse:
        amp = eng.amplitude.output("signal").numpy()
        assert np.max(np.abs(amp - ref["amplitude"])) <= AMPLITUDE_TOL_DB * 100
        assert_bit_equal(eng.fft.output("signal").numpy(), ref["fft"], "fft is exact in every mode")
    # integer bin work is exact given its float input ...
    bins = np.zeros(n * h, np.float32)
    oracle.spectrogram(bins, got, h)
    oracle.spectrogram(bins, got, h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins)
    # ... and differs from the all-CPU chain only where a value sits within an ulp of a bin edge
    ref_bins = np.zeros(n * h, np.float32)
    oracle.spectrogram(ref_bins, ref["range"], h)
    oracle.spectrogram(ref_bins, ref["range"], h)
    assert np.mean(bins != ref_bins) < 1e-4


def test_fast_modules_edge_values(js, oracle):
    x = np.array([-np.inf, -400.0, -100.0, -50.0, -49.999, 0.0, 50.0, np.inf, np.nan], np.float32)
    m = js.Module("range", {"min": -100.0, "max": 0.0}, {"signal": js.Tensor.from_numpy(x)},
                  provider="fast")
    rt = js.Runtime([m])
    rt.compute()
    got, ref = m.output("signal").numpy(), oracle.range_(x, -100.0, 0.0)
    assert np.isnan(got[-1]) and got[0] == 0.0 and got[-2] == 1.0
    assert np.max(np.abs(got[:-1] - ref[:-1])) <= RANGE_TOL_ABS
    z = np.zeros(8, np.complex64)
    a = js.Module("amplitude", {}, {"signal": js.Tensor.from_numpy(z)}, provider="fast")
    rt = js.Runtime([a])
    rt.compute()
    assert np.all(np.isneginf(a.output("signal").numpy()))  # exact zero -> -inf in every mode


def test_mixed_providers_are_not_fused(js):
    x = csignal(np.random.default_rng(0), (4, 1024))
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src)
    fast_range = js.Module("range", {"min": -100.0, "max": 0.0},
                           {"signal": eng.amplitude.output("signal")}, "fast_range", provider="fast")
    rt = js.Runtime(eng.modules[:-1] + [fast_range], fuse=True)
    assert not any("fast_range" in u and u.startswith("spectrum_fused") for u in rt.units)
    rt.compute()
