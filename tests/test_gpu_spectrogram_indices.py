"""The Spectrogram fed with one-byte row indices by the fused spectrum unit (fft_side.hip: StoreAmplitudeRangeSideT,
spectrogram.hip: spectrogram_index_kernel): the planner's decision, and the state bit for bit against the oracle
(spectrogram/module_impl_native_cpu.cc:61-87) over shapes that exercise every branch of the index kernel -- rows that do
not fill a round of 1024, more than one round, heights that are not powers of two, saturating and empty columns."""
import numpy as np
import pytest

from test_gpu_chain import tone_batch
from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def _chain(js, x, h, provider="generic", **runtime):
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime(eng.modules + [spec], fuse=True, **runtime)
    return eng, spec, rt


def _fed(rt):
    return any(u.startswith("spectrum_fused(") and u.endswith("+indices") for u in rt.units)


@pytest.mark.parametrize("n,b,h", [(4096, 64, 256), (4096, 5, 256), (1024, 1000, 100), (2048, 1500, 255),
                                   (8192, 17, 2), (4096, 2100, 37), (1024, 1024, 256)])
@pytest.mark.parametrize("graph", [False, True])
def test_index_fed_spectrogram_matches_the_oracle(js, oracle, n, b, h, graph):
    x = tone_batch(oracle, b, n, 100 + b)
    x[:, :] *= np.float32(0.5)
    x[::3] *= np.float32(30.0)       # some rows saturate the top of the range, some columns never hit
    eng, spec, rt = _chain(js, x, h, graph=graph)
    assert _fed(rt), rt.units
    cycles = 4
    rt.compute(cycles)
    ref = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    assert_bit_equal(eng.buffer.numpy(), ref, "the F32 output is still written")
    bins = np.zeros(n * h, np.float32)
    for _ in range(cycles):
        oracle.spectrogram(bins, ref, h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"state, n={n} b={b} h={h}")


def test_index_fed_over_many_cycles_and_span_graphs(js, oracle):
    """A ring of 3 slots (period 3), 23 cycles in calls of odd lengths: period graphs, spans and eager cycles."""
    n, b, h, slots = 4096, 48, 256, 3
    xs = [tone_batch(oracle, b, n, 7 + s) * np.float32(0.2 + 0.4 * s) for s in range(slots)]
    ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "ring")
    buf = ring.output("buffer")
    for s in range(slots):
        buf.ring_select(s).copy_from(xs[s])
    buf.ring_select(0)
    eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime([ring] + eng.modules + [spec], fuse=True, graph=True)
    assert _fed(rt), rt.units
    bins = np.zeros(n * h, np.float32)
    refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in xs]
    done = 0
    for call in (1, 3, 4, 2, 7, 6):
        rt.compute(call)
        for k in range(call):
            oracle.spectrogram(bins, refs[(done + k) % slots], h)
        done += call
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"after {done} cycles")
    assert done == 23


def test_fast_provider_indices_equal_the_exact_bins(js, oracle):
    n, b, h = 4096, 96, 256
    x = tone_batch(oracle, b, n, 31)
    eng, spec, rt = _chain(js, x, h, provider="fast", graph=True)
    assert _fed(rt), rt.units
    rt.compute(3)
    ref = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    assert np.max(np.abs(eng.buffer.numpy() - ref)) <= 4e-7
    bins = np.zeros(n * h, np.float32)
    for _ in range(3):
        oracle.spectrogram(bins, ref, h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, "fast provider + bin guard + indices")


@pytest.mark.parametrize("fmt,np_t,scale", [("CI16", np.int16, 32768.0), ("CI8", np.int8, 128.0), ("CU8", np.uint8, 128.0)])
def test_raw_sample_input_with_indices(js, oracle, fmt, np_t, scale):
    """Cast folded into the first load AND the indices as a side output: raw samples in, values + indices out."""
    n, b, h = 4096, 40, 256
    rng = np.random.default_rng(5)
    info = np.iinfo(np_t)
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": 2, "live": True, "dtype": fmt}, {}, "sdr")
    eng = js.SpectrumEngine(src.output("buffer"), enable_scale=True, range_min=-100.0, range_max=0.0)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime([src] + eng.modules + [spec], fuse=True, graph=True)
    assert _fed(rt) and any("cast_input" in u and u.startswith("spectrum_fused(") for u in rt.units), rt.units
    bins = np.zeros(n * h, np.float32)
    for k in range(3):
        raw = (rng.integers(info.min, info.max + 1, (b, n, 2)) // (1 + 7 * k)).astype(np_t)
        assert src.ring_push(raw.reshape(-1, 2)) == "success"
        assert rt.compute(1) == "success"
        ref = oracle.spectrum_chain(oracle.cast(raw, complex_pairs=True), -100.0, 0.0)["range"]
        assert_bit_equal(eng.buffer.numpy(), ref, f"{fmt} values, batch {k}")
        oracle.spectrogram(bins, ref, h)
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"{fmt} state, batch {k}")


def test_planner_keeps_the_value_path_when_it_must(js, oracle):
    n, b = 4096, 16
    x = csignal(np.random.default_rng(3), (b, n))
    # height > 256: an index does not fit a byte
    _, spec, rt = _chain(js, x, 512)
    assert not _fed(rt) and any(u.startswith("spectrum_fused(") for u in rt.units)
    rt.compute(2)
    ref = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    bins = np.zeros(n * 512, np.float32)
    oracle.spectrogram(bins, ref, 512)
    oracle.spectrogram(bins, ref, 512)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins)
    # two Spectrograms on one output: both read the values
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0)
    s1 = js.Module("spectrogram", {"height": 256}, {"signal": eng.buffer}, "s1")
    s2 = js.Module("spectrogram", {"height": 128}, {"signal": eng.buffer}, "s2")
    rt = js.Runtime(eng.modules + [s1, s2], fuse=True)
    assert not _fed(rt)
    rt.compute(1)
    for s, h in ((s1, 256), (s2, 128)):
        bins = np.zeros(n * h, np.float32)
        oracle.spectrogram(bins, ref, h)
        assert_bit_equal(s.state("frequencyBins").numpy().reshape(-1), bins)
    # a waterfall beside the one Spectrogram does not matter: it reads the values, the Spectrogram the indices
    eng = js.SpectrumEngine(js.Tensor.from_numpy(x, sample=1, batch=0), enable_scale=True, range_min=-100.0, range_max=0.0)
    s1 = js.Module("spectrogram", {"height": 256}, {"signal": eng.buffer}, "s1")
    wf = js.Module("waterfall", {"height": 64}, {"signal": eng.buffer}, "wf")
    rt = js.Runtime(eng.modules + [s1, wf], fuse=True)
    assert _fed(rt), rt.units
    rt.compute(1)
    bins = np.zeros(n * 256, np.float32)
    oracle.spectrogram(bins, ref, 256)
    assert_bit_equal(s1.state("frequencyBins").numpy().reshape(-1), bins)
    assert_bit_equal(wf.state("frequencyBins").numpy().reshape(64, n)[:b], ref)
    # unfused runtime: no side output to read
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0)
    s1 = js.Module("spectrogram", {"height": 256}, {"signal": eng.buffer}, "s1")
    rt = js.Runtime(eng.modules + [s1], fuse=False)
    assert not _fed(rt)
    rt.compute(1)
    assert_bit_equal(s1.state("frequencyBins").numpy().reshape(-1), bins)
