"""The spectrum chain (Window -> Invert -> Reshape -> Multiply -> FFT -> Amplitude -> Range ->
Spectrogram) on the GPU vs the CPU oracle: module-by-module, fused, under hipGraph replay, and at
BASELINE config sizes through size-independent properties."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["pipe", "slot", "quad"])
def fft_kernel_variant(request, switch):
    """Every test here runs against the FFT kernel variants (pipelined / slot / quad: the round-5 4096-point side kernel,
    the default when the variable is unset), same bits."""
    switch("JST_FFT_KERNEL", request.param)
    yield


def tone_batch(oracle, b, n, seed, sigma=1e-3):
    """SURVEY 8(d) C2 input: row r = CW tone at bin 100.25 + r (signal_generator arithmetic) +
    complex AWGN sigma from default_rng(seed)."""
    rng = np.random.default_rng(seed)
    fs = 2.0e6
    x = np.empty((b, n), np.complex64)
    for r in range(b):
        x[r], _ = oracle.signal_cosine(n, 1.0, (100.25 + r) * fs / n, fs)
    noise = rng.standard_normal((b, n, 2)).astype(np.float32) * np.float32(sigma)
    return (x + (noise[..., 0] + 1j * noise[..., 1])).astype(np.complex64)


def build(js, x, h=256, fuse=True, graph=True, scale=True, timing=False, pipeline=False):
    src = js.Tensor.from_numpy(x, sample=x.ndim - 1, **({"batch": 0} if x.ndim > 1 else {}))
    eng = js.SpectrumEngine(src, enable_scale=scale, range_min=-100.0, range_max=0.0)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime(eng.modules + [spec], graph=graph, fuse=fuse, timing=timing, pipeline=pipeline)
    return src, eng, spec, rt


@pytest.mark.parametrize("n,b", [(4096, 1), (4096, 16), (1024, 33), (256, 5), (8192, 3)])
def test_unfused_every_stage_bit_exact(js, oracle, n, b):
    x = tone_batch(oracle, b, n, 1234)
    src, eng, spec, rt = build(js, x, fuse=False, graph=False)
    assert not any(u.startswith("spectrum_fused") for u in rt.units)
    rt.compute()
    ref = oracle.spectrum_chain(x, -100.0, 0.0)
    assert_bit_equal(eng.invert.output("signal").numpy(), ref["window"], "invert(window)")
    assert_bit_equal(eng.multiply.output("product").numpy(), ref["product"], "multiply")
    assert_bit_equal(eng.fft.output("signal").numpy(), ref["fft"], "fft")
    assert_bit_equal(eng.amplitude.output("signal").numpy(), ref["amplitude"], "amplitude")
    assert_bit_equal(eng.buffer.numpy(), ref["range"], "range")
    assert eng.buffer.axes == {"sample": 1, "batch": 0, "channel": None}


@pytest.mark.parametrize("n,b", [(4096, 1), (4096, 16), (1024, 33), (256, 5), (16384, 2)])
@pytest.mark.parametrize("scale", [True, False])
def test_fused_equals_oracle_and_unfused(js, oracle, n, b, scale):
    x = tone_batch(oracle, b, n, 99)
    src, eng, spec, rt = build(js, x, fuse=True, graph=False, scale=scale)
    assert any(u.startswith("spectrum_fused") for u in rt.units), rt.units
    rt.compute()
    ref = oracle.spectrum_chain(x, -100.0, 0.0)
    assert_bit_equal(eng.buffer.numpy(), ref["range" if scale else "amplitude"], "fused output")


@pytest.mark.parametrize("pipeline", [False, True])
def test_spectrogram_state_over_cycles_with_graph(js, oracle, pipeline):
    n, b, h = 4096, 32, 256
    x = tone_batch(oracle, b, n, 1234)
    src, eng, spec, rt = build(js, x, h=h, fuse=True, graph=True, pipeline=pipeline)
    ref_out = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    ref_bins = np.zeros(n * h, np.float32)
    for cycle in range(1, 5):
        rt.compute()
        oracle.spectrogram(ref_bins, ref_out, h)
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), ref_bins, f"cycle {cycle}")
    if pipeline:  # producer and surface lanes are separate graphs on two streams; several periods in flight
        assert rt.period == 1 and rt.graph_active
        rt.compute(4)
        for _ in range(4):
            oracle.spectrogram(ref_bins, ref_out, h)
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), ref_bins, "pipelined replays")
    assert rt.graph_active
    # static modules settled after the first cycle (scheduler_synchronous.cc:534-546)
    assert eng.window.timing["cycles"] == 1 and eng.invert.timing["cycles"] == 1
    done = 8 if pipeline else 4
    assert eng.fft.timing["cycles"] == done and spec.timing["cycles"] == done
    # new data through the SAME graph: the captured pointers stay valid, contents change
    x2 = tone_batch(oracle, b, n, 77)
    src.copy_from(x2)
    rt.compute()
    ref2 = oracle.spectrum_chain(x2, -100.0, 0.0)["range"]
    oracle.spectrogram(ref_bins, ref2, h)
    assert_bit_equal(eng.buffer.numpy(), ref2, "second input")
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), ref_bins, "bins after new input")


@pytest.mark.parametrize("pipeline,timing", [(False, False), (True, False), (False, True)])
def test_ring_source_period_and_graph(js, oracle, pipeline, timing):
    n, b, h, slots = 1024, 8, 64, 4
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "source")
    out = src.output("buffer")
    data = [tone_batch(oracle, b, n, 500 + s) for s in range(slots)]
    for s in range(slots):
        out.ring_select(s).copy_from(data[s])
    out.ring_select(0)
    assert out.axes == {"sample": 1, "batch": 0, "channel": None}  # soapy/module_impl.cc:197-201
    eng = js.SpectrumEngine(out)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    water = js.Module("waterfall", {"height": 16}, {"signal": eng.buffer}, "waterfall")
    # timing=True: one cycle of every sixteenth period runs eagerly between real event records, the rest as span graphs
    rt = js.Runtime([src] + eng.modules + [spec, water], graph=True, fuse=True, pipeline=pipeline, timing=timing)
    assert rt.period == slots
    ring, wstate = np.zeros((16, n), np.float32), (0, 0)
    refs = [oracle.spectrum_chain(d, -100.0, 0.0)["range"] for d in data]
    bins = np.zeros(n * h, np.float32)
    total = 0
    # whole-period replays mixed with heads and tails that do not fill a period: those replay as span graphs
    # (captured per (phase, length), replayed from the cache the second time: the source's host cursor must follow)
    for chunk in (1, 3, 4, 8, 2, 4, 3, 3, 5, 2, 4, 7, 3, 3, 16, 9, 40):
        rt.compute(chunk)
        for _ in range(chunk):
            oracle.spectrogram(bins, refs[total % slots], h)
            wstate = oracle.waterfall(ring, wstate, refs[total % slots], 16)
            total += 1
        assert_bit_equal(water.state("frequencyBins").numpy(), ring, f"waterfall after {total}")
        assert_bit_equal(eng.buffer.numpy(), refs[(total - 1) % slots], f"after {total} cycles")
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"bins after {total}")
    assert rt.graph_active
    if timing:
        assert 0.0 < rt.unit_mean_ms("spectrum_fused") < 5.0 and spec.timing["cycles"] == total


def test_full_size_properties(js, oracle):
    """BASELINE config 2 size (1024 x 4096): the oracle takes a while on the full batch, so check a
    row sample bit-exactly plus size-independent properties on everything."""
    n, b, h = 4096, 1024, 256
    x = tone_batch(oracle, 8, n, 1234)
    big = np.tile(x, (b // 8, 1))
    src, eng, spec, rt = build(js, big, h=h, fuse=True, graph=True)
    rt.compute(2)
    out = eng.buffer.numpy()
    ref8 = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    assert_bit_equal(out[:8], ref8, "first rows")
    assert_bit_equal(out[-8:], ref8, "last rows")
    assert np.array_equal(out.reshape(b // 8, 8, n), np.broadcast_to(out[:8], (b // 8, 8, n)))
    # peak of row r sits at the centred bin n/2 + 100 + r (invert = fftshift by modulation)
    assert [int(np.argmax(out[r])) for r in range(8)] == [n // 2 + 100 + r for r in range(8)]
    bins = spec.state("frequencyBins").numpy().reshape(h, n)
    assert np.all(bins[0] == 0) and bins.max() <= 1.0 and bins.min() >= 0.0  # row 0 excluded
    ref_bins = np.zeros(n * h, np.float32)
    tiled = np.tile(ref8, (b // 8, 1))
    oracle.spectrogram(ref_bins, tiled, h)
    oracle.spectrogram(ref_bins, tiled, h)
    assert_bit_equal(bins.reshape(-1), ref_bins, "full-size spectrogram")


def test_c2_literal_full_batch_three_cycles(js, oracle):
    """BASELINE configs[1] exactly as SURVEY 8(d) C2 writes it: CF32[1024, 4096], row b = unit tone at bin
    100.25 + b (signal_generator arithmetic) + complex AWGN sigma 1e-3 from default_rng(1234), 1024 DISTINCT
    rows, window 4096, range -100..0 dB, spectrogram height 256, three cycles under graph replay.  The range
    output of the WHOLE batch and the whole spectrogram state are compared bit for bit with the oracle's dense
    chain pass (oracle/jst_oracle.c: jst_oracle_chain_pass, ~50 MS/s) after every cycle."""
    n, b, h = 4096, 1024, 256
    x = tone_batch(oracle, b, n, 1234)
    src, eng, spec, rt = build(js, x, h=h, fuse=True, graph=True)
    assert any(u.startswith("spectrum_fused") for u in rt.units), rt.units
    bins = np.zeros(n * h, np.float32)
    for cycle in range(1, 4):
        rt.compute(1)
        ref = oracle.chain_pass(x, bins, h)
        assert_bit_equal(eng.buffer.numpy(), ref, f"range output, cycle {cycle}")
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"spectrogram state, cycle {cycle}")
    assert rt.graph_active
    # every row's peak sits at the centred bin n/2 + 100 + b (mod n): no row was dropped or duplicated
    peaks = np.argmax(eng.buffer.numpy(), axis=1)
    assert np.array_equal(peaks, (n // 2 + 100 + np.arange(b)) % n)
    rt.destroy()


def test_c2_bench_ring_two_graph_periods_full_size(js, oracle):
    """The data bench.py times: its 16-slot ring of synth_slot() batches (1024 x 4096 each, 512 MiB), the runtime
    built like bench.py builds it (ring_source, fused, graph, TIMING), run through two whole graph periods plus a
    tail; after every chunk the last cycle's range output (all 1024 rows) and the spectrogram state are bit-equal
    to the oracle's.  This is the parity the bench line's `parity` stamp re-checks on its own run."""
    import bench
    n, b, h, slots = bench.N_FFT, bench.BATCHES, bench.HEIGHT, 16
    source = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "source")
    buf = source.output("buffer")
    rng = np.random.default_rng(1234)
    data = [bench.synth_slot(rng, s) for s in range(slots)]
    for s in range(slots):
        buf.ring_select(s).copy_from(data[s])
    buf.ring_select(0)
    eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime([source] + eng.modules + [spec], graph=True, fuse=True, timing=True)
    assert rt.period == slots
    scratch = np.zeros(n * h, np.float32)
    refs = [oracle.chain_pass(d, scratch.copy(), h) for d in data]   # range output per slot (state-free)
    bins = np.zeros(n * h, np.float32)
    total = 0
    for chunk in (1, 15, 16, 16, 5):   # settle cycle, rest of period 0, two whole periods, a tail span
        rt.compute(chunk)
        for _ in range(chunk):
            oracle.spectrogram(bins, refs[total % slots], h)
            total += 1
        assert_bit_equal(eng.buffer.numpy(), refs[(total - 1) % slots], f"range output after {total} cycles")
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"state after {total} cycles")
    assert rt.graph_active and 0.0 < rt.unit_mean_ms("spectrum_fused") < 5.0
    rt.destroy()


def test_golden_fixture(js):
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "spectrum_chain_c1.npz")
    g = np.load(path)
    src, eng, spec, rt = build(js, g["x"], h=int(g["height"]), fuse=True, graph=False)
    rt.compute(int(g["cycles"]))
    assert_bit_equal(eng.buffer.numpy(), g["range"], "golden range")
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), g["bins"], "golden bins")
