"""Cycle batching (JST_RUNTIME_BATCH, Runtime::planBatch): with a resident ring source the cycles of a captured ring
period -- and of every span of it -- run as ONE launch per unit (the persistent fused spectrum kernel over all the span's
slots, the index-fed Spectrogram over all its index tensors with the state tile in registers).  What is visible after
every compute() must be what the per-cycle submissions leave, bit for bit: every cycle's range output (in its ring slot),
the handle showing the latest cycle, and the Spectrogram state (spectrogram/module_impl_native_cpu.cc:61-87 applied once
per cycle) -- against the oracle and against the same chain run per cycle."""
import numpy as np
import pytest

from test_gpu_chain import tone_batch
from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def _ring_chain(js, xs, h, provider="generic", dtype=None, **runtime):
    b, n = xs[0].shape[:2]
    cfg = {"batches": b, "samples": n, "slots": len(xs)}
    if dtype:
        cfg["dtype"] = dtype
    ring = js.Module("ring_source", cfg, {}, "ring")
    buf = ring.output("buffer")
    for s, x in enumerate(xs):
        buf.ring_select(s).copy_from(x)
    buf.ring_select(0)
    eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    mods = [ring] + eng.modules + [spec]
    rt = js.Runtime(mods, fuse=True, graph=True, **runtime)
    rt._keep = mods
    return eng, spec, rt


@pytest.mark.parametrize("n,b,h,slots", [(4096, 48, 256, 3), (1024, 130, 100, 5), (2048, 1100, 255, 2), (4096, 8, 37, 16)])
def test_batched_runtime_matches_the_oracle_cycle_by_cycle(js, oracle, n, b, h, slots):
    xs = [tone_batch(oracle, b, n, 11 + 3 * s) * np.float32(0.15 + 0.35 * s) for s in range(slots)]
    for x in xs:
        x[::3] *= np.float32(20.0)
    eng, spec, rt = _ring_chain(js, xs, h, batch=True)
    assert rt.batched, rt.units
    refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in xs]
    bins = np.zeros(n * h, np.float32)
    done = 0
    # eager settle, whole periods, spans from every phase, a period and a tail, several periods in one call
    for call in (1, slots, 1, slots - 1, 2 * slots + 1, slots + slots // 2 + 1, 3 * slots, 2):
        rt.compute(call)
        for k in range(call):
            oracle.spectrogram(bins, refs[(done + k) % slots], h)
        done += call
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"state after {done} cycles (call of {call})")
        assert_bit_equal(eng.buffer.numpy(), refs[(done - 1) % slots], f"the handle shows cycle {done - 1}'s output")
        if call >= slots:  # every slot of the output ring was written by this call: each holds its cycle's output
            latest = (done - 1) % slots
            for s in range(slots):
                assert_bit_equal(eng.buffer.ring_select(s).numpy(), refs[s], f"output ring slot {s} after {done} cycles")
            eng.buffer.ring_select(latest)


@pytest.mark.parametrize("provider", ["generic", "fast"])
def test_batched_equals_per_cycle_bit_for_bit(js, oracle, provider):
    """Same data, same call pattern, batch on / off: outputs and state identical whatever the provider."""
    n, b, h, slots = 4096, 64, 256, 4
    xs = [tone_batch(oracle, b, n, 5 + s) * np.float32(0.3 + 0.2 * s) for s in range(slots)]
    got = []
    for batch in (False, True):
        eng, spec, rt = _ring_chain(js, xs, h, provider=provider, batch=batch)
        assert rt.batched == batch
        trace = []
        for call in (1, 4, 3, 9, 2, 8):
            rt.compute(call)
            trace.append((eng.buffer.numpy().copy(), spec.state("frequencyBins").numpy().copy()))
        got.append(trace)
        rt.destroy()
    for i, ((o0, s0), (o1, s1)) in enumerate(zip(*got)):
        assert_bit_equal(o1, o0, f"output after call {i}")
        assert_bit_equal(s1, s0, f"spectrogram state after call {i}")


def test_batched_raw_sample_ring(js, oracle):
    """CI16 ring: the cast folded into the batched launch's first load."""
    n, b, h, slots = 4096, 24, 256, 3
    rng = np.random.default_rng(9)
    raws = [(rng.integers(-32768, 32768, (b, n, 2)) // (1 + 5 * s)).astype(np.int16) for s in range(slots)]
    eng, spec, rt = _ring_chain(js, raws, h, dtype="CI16", batch=True)
    assert rt.batched and any("cast_input" in u and u.startswith("spectrum_fused(") for u in rt.units), rt.units
    refs = [oracle.spectrum_chain(oracle.cast(r, complex_pairs=True), -100.0, 0.0)["range"] for r in raws]
    bins = np.zeros(n * h, np.float32)
    done = 0
    for call in (1, 3, 5, 6):
        rt.compute(call)
        for k in range(call):
            oracle.spectrogram(bins, refs[(done + k) % slots], h)
        done += call
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"state after {done} cycles")
        assert_bit_equal(eng.buffer.numpy(), refs[(done - 1) % slots], f"output of cycle {done - 1}")


def test_planner_stays_per_cycle_when_a_unit_cannot_batch(js, oracle):
    n, b, h, slots = 4096, 16, 256, 3
    xs = [tone_batch(oracle, b, n, 21 + s) for s in range(slots)]
    # height > 256: the Spectrogram reads values, not indices -> no span support in the spectrum unit -> per cycle
    eng, spec, rt = _ring_chain(js, xs, 512, batch=True)
    assert not rt.batched
    rt.compute(7)
    refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in xs]
    bins = np.zeros(n * 512, np.float32)
    for k in range(7):
        oracle.spectrogram(bins, refs[k % slots], 512)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins)
    assert_bit_equal(eng.buffer.numpy(), refs[6 % slots])
    rt.destroy()
    # batch=False: per cycle; no flag: graph + fuse batch whenever the chain allows it
    eng, spec, rt = _ring_chain(js, xs, h, batch=False)
    assert not rt.batched
    rt.destroy()
    eng, spec, rt = _ring_chain(js, xs, h)
    assert rt.batched
    rt.destroy()
    eng, spec, rt = _ring_chain(js, xs, 512)
    assert not rt.batched


def test_sink_surfaces_ride_inside_batched_spans(js, oracle):
    """A waterfall and a lineplot on the same range output have no span form: in a batched runtime they run their n
    per-cycle submissions behind the span's big launches, each on its cycle's slot of the output ring.  Every state must
    equal the per-cycle runtime's, bit for bit, after every call."""
    n, b, h, slots = 4096, 24, 256, 4
    xs = [tone_batch(oracle, b, n, 40 + s) * np.float32(0.2 + 0.25 * s) for s in range(slots)]
    traces = []
    for batch in (False, True):
        ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "ring")
        buf = ring.output("buffer")
        for s, x in enumerate(xs):
            buf.ring_select(s).copy_from(x)
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
        spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
        wf = js.Module("waterfall", {"height": 100}, {"signal": eng.buffer}, "waterfall")
        lp = js.Module("lineplot", {"averaging": 4}, {"signal": eng.buffer}, "lineplot")
        rt = js.Runtime([ring] + eng.modules + [spec, wf, lp], fuse=True, graph=True, batch=batch)
        assert rt.batched == batch, rt.units
        trace = []
        for call in (1, 4, 3, 6, 2, 9, 8):
            rt.compute(call)
            trace.append([spec.state("frequencyBins").numpy().copy(), wf.state("frequencyBins").numpy().copy(),
                          wf.state("ringState").numpy().copy(), lp.state("signalPoints").numpy().copy(),
                          lp.state("averagingBuffer").numpy().copy(), eng.buffer.numpy().copy()])
        traces.append(trace)
        rt.destroy()
    names = ("spectrogram", "waterfall bins", "waterfall cursor", "lineplot points", "lineplot average", "range output")
    for i, (per_cycle, batched) in enumerate(zip(*traces)):
        for name, a, bb in zip(names, per_cycle, batched):
            if a.dtype == np.float32:
                assert_bit_equal(bb, a, f"{name} after call {i}")
            else:
                assert np.array_equal(bb, a), f"{name} after call {i}"


def test_batched_timing_samples_cover_a_period(js, oracle):
    n, b, h, slots = 4096, 32, 256, 4
    xs = [tone_batch(oracle, b, n, 2 + s) for s in range(slots)]
    eng, spec, rt = _ring_chain(js, xs, h, batch=True, timing=True)
    assert rt.batched
    rt.compute(1)
    rt.compute(slots - 1)
    rt.reset_timing()
    rt.compute(slots * 40)
    fused = next(u for u in rt.units if u.startswith("spectrum_fused"))
    assert rt.unit_mean_ms(fused) > 0
    assert rt.unit_mean_cycles(fused) == slots
    assert rt.unit_mean_cycles("spectrogram") == slots
    refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in xs]
    bins = np.zeros(n * h, np.float32)
    for k in range(slots * 41):
        oracle.spectrogram(bins, refs[k % slots], h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, "timed (eager batched) and replayed periods")


@pytest.mark.parametrize("provider", ["fast", "generic"])
def test_side_kernel_with_a_complex_operand(js, oracle, provider):
    """Provider "fast" multiplies a REAL window as two products per sample (fft_lds.hh: REALFAST, decided per wavefront
    on the device); an operand with imaginary parts must take std::complex's full product.  A complex broadcast operand
    through multiply -> fft -> amplitude -> range -> index-fed Spectrogram: values and state against the oracle."""
    n, b, h = 4096, 40, 256
    rng = np.random.default_rng(17)
    x = tone_batch(oracle, b, n, 3) * np.float32(0.4)
    w = (rng.standard_normal((1, n, 2), dtype=np.float32) * np.float32(0.5)).view(np.complex64)[..., 0]
    sig = js.Tensor.from_numpy(x, sample=1, batch=0)
    win = js.Tensor.from_numpy(w, sample=1)
    mul = js.Module("multiply", {}, {"a": sig, "b": win}, "multiply")
    fft = js.Module("fft", {"forward": True}, {"signal": mul.output("product")}, "fft")
    amp = js.Module("amplitude", {}, {"signal": fft.output("signal")}, "amplitude", provider=provider)
    rge = js.Module("range", {"min": -100.0, "max": 0.0}, {"signal": amp.output("signal")}, "range", provider=provider)
    spec = js.Module("spectrogram", {"height": h}, {"signal": rge.output("signal")}, "spectrogram")
    rt = js.Runtime([mul, fft, amp, rge, spec], fuse=True, graph=True)
    assert any(u.startswith("spectrum_fused(") and u.endswith("+indices") for u in rt.units), rt.units
    rt.compute(3)
    ref = oracle.range_(oracle.amplitude(oracle.fft_c2c(oracle.multiply(x, w)), n), -100.0, 0.0)
    got = rge.output("signal").numpy()
    if provider == "generic":
        assert_bit_equal(got, ref, "complex operand, exact provider")
    else:
        assert np.max(np.abs(got - ref)) <= 4e-7
    bins = np.zeros(n * h, np.float32)
    for _ in range(3):
        oracle.spectrogram(bins, ref, h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, "state with a complex operand")


def test_runtime_recreated_on_the_same_modules_batches_again(js, oracle):
    """The planner turned the range output and the row indices into rings; a second runtime over the same modules must
    find them usable (not fall back to the value-fed, per-cycle path) and carry the Spectrogram state on."""
    n, b, h, slots = 4096, 16, 256, 3
    xs = [tone_batch(oracle, b, n, 60 + s) for s in range(slots)]
    eng, spec, rt = _ring_chain(js, xs, h, batch=True)
    mods = rt._keep
    assert rt.batched
    rt.compute(5)
    rt.destroy()
    rt2 = js.Runtime(mods, fuse=True, graph=True, batch=True)
    assert rt2.batched and any(u.endswith("+indices") for u in rt2.units), rt2.units
    rt2.compute(7)
    refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in xs]
    bins = np.zeros(n * h, np.float32)
    for k in range(12):  # the ring source's cursor went on from where the first runtime left it
        oracle.spectrogram(bins, refs[k % slots], h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, "state across two runtimes")
    assert_bit_equal(eng.buffer.numpy(), refs[11 % slots])


@pytest.mark.parametrize("n,b", [(65536, 4), (12000, 6)])
def test_tiled_spectrum_unit_batches_too(js, oracle, n, b):
    """The LDS-tiled spectrum unit (65536 points: config 5; a mixed-radix length) has a span form as well: runs of
    consecutive ring slots as one columns / blocks launch pair, a lineplot riding behind it as a sink.  Batched and
    per-cycle runtimes must agree bit for bit after every call; the first cycle is checked against the oracle."""
    slots = 3
    rng = np.random.default_rng(n)
    xs = [((rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))) * (0.1 + 0.2 * s)).astype(np.complex64)
          for s in range(slots)]
    traces = []
    for batch in (False, True):
        ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "ring")
        buf = ring.output("buffer")
        for s, x in enumerate(xs):
            buf.ring_select(s).copy_from(x)
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
        lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
        rt = js.Runtime([ring] + eng.modules + [lp], fuse=True, graph=True, batch=batch)
        assert rt.batched == batch, rt.units
        trace = []
        for call in (1, 3, 2, 5, 7):
            rt.compute(call)
            trace.append([eng.buffer.numpy().copy(), lp.state("averagingBuffer").numpy().copy(),
                          lp.state("signalPoints").numpy().copy()])
        traces.append(trace)
        if batch:
            for s in range(slots):
                assert_bit_equal(eng.buffer.ring_select(s).numpy(), oracle.spectrum_chain(xs[s], -100.0, 0.0)["range"],
                                 f"output ring slot {s} against the oracle")
        rt.destroy()
    for i, (per_cycle, batched) in enumerate(zip(*traces)):
        for name, a, bb in zip(("range output", "lineplot average", "lineplot points"), per_cycle, batched):
            assert_bit_equal(bb, a, f"{name} after call {i}")
