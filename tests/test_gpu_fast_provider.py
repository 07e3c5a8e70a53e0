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
    ref_bins = np.zeros(n * h, np.float32)
    oracle.spectrogram(ref_bins, ref["range"], h)
    oracle.spectrogram(ref_bins, ref["range"], h)
    if fuse:
        # ... and, fused, equals the all-CPU chain bin for bin: the epilogue recomputes with the exact
        # arithmetic whatever falls within the fast path's error of a bin edge (dev::BinGuard)
        assert_bit_equal(bins, ref_bins, "spectrogram bins, fast provider with bin guard")
    else:  # module by module the range module cannot know its consumer: edge cases may move a bin
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


def _bins(r, h):
    """The Spectrogram bin rule (spectrogram.hip): hit <=> 1 <= f < h with f = r * h in F32; -1 = no hit."""
    f = r.astype(np.float32) * np.float32(h)
    with np.errstate(invalid="ignore"):
        hit = (f >= 1.0) & (f < h)
    return np.where(hit, f.astype(np.int64, casting="unsafe"), -1)


@pytest.mark.parametrize("h0,h1", [(256.0, 0.0), (100.0, 0.0), (2048.0, 256.0), (17.0, 1000.0)])
def test_bin_guard_makes_fast_bins_exact_on_dense_sweeps(js, h0, h1):
    """The fused fast epilogue against the exact one on 2^24 arbitrary spectrum values per case (magnitudes
    log-uniform over the whole range window and beyond, zeros, huge and tiny values): every bin equal for
    both guarded heights, floats within 3e-7; and the same sweep WITHOUT the guard does move bins, i.e.
    the sweep is dense enough to see the effect the guard removes."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(int(h0) * 7 + int(h1))
    n = 1 << 24
    mag = 10.0 ** rng.uniform(-7.5, 1.5, n)
    ph = rng.uniform(0, 2 * np.pi, n)
    v = (mag * np.exp(1j * ph)).astype(np.complex64)
    v[:1000] = 0
    v[1000:2000] *= 1e20
    v[2000:3000] *= 1e-30
    x = torch.from_numpy(v.view(np.float32).copy()).cuda()
    exact = torch.empty(n, dtype=torch.float32, device="cuda")
    fast = torch.empty_like(exact)
    coeff = float(np.float32(20.0 * np.log10(1.0 / 4096.0)))
    lo, hi = -100.0, 0.0
    scale, offset = float(np.float32(1.0 / (hi - lo))), float(np.float32(-lo / (hi - lo)))

    def run(g0, g1):
        r = js._lib.jst_probe_amplitude_range(C.c_void_p(x.data_ptr()), C.c_void_p(exact.data_ptr()),
                                              C.c_void_p(fast.data_ptr()), n, coeff, scale, offset, g0, g1)
        assert r == 0, js._lib.jst_last_error()
        return exact.cpu().numpy(), fast.cpu().numpy()

    e, f = run(h0, h1)
    assert np.max(np.abs(e - f)) <= RANGE_TOL_ABS
    print("max |fast - exact| =", float(np.max(np.abs(e - f))))
    for h in (h0, h1):
        if h:
            assert np.array_equal(_bins(e, h), _bins(f, h)), h
    e2, f2 = run(0.0, 0.0)  # unguarded: same floats tolerance, but some bins move
    assert np.max(np.abs(e2 - f2)) <= RANGE_TOL_ABS
    moved = int(np.sum(_bins(e2, h0) != _bins(f2, h0)))
    assert moved > 0, "sweep too sparse to exercise the guard"
    assert moved < 1e-3 * n
