"""Runtime / scheduler behaviour on the device: execution order, static settlement, fusion
planning, graph replay == eager, per-unit timing (src/scheduler_synchronous.cc:534-546,574-696;
src/runtime/native/cuda/impl.cc:185-272)."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def test_order_is_topological_regardless_of_insertion(js):
    x = csignal(np.random.default_rng(1), (4, 256))
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src)
    shuffled = list(reversed(eng.modules))
    rt = js.Runtime(shuffled)
    pos = {name: i for i, name in enumerate(rt.order)}
    assert pos["spectrum.window"] < pos["spectrum.invert"] < pos["spectrum.multiply"]
    assert pos["spectrum.multiply"] < pos["spectrum.fft"] < pos["spectrum.amplitude"] < pos["spectrum.range"]
    rt.compute()
    rt2 = js.Runtime(js.SpectrumEngine(src).modules)
    rt2.compute()


def test_fusion_is_refused_when_an_intermediate_has_another_consumer(js, oracle):
    x = csignal(np.random.default_rng(2), (4, 1024))
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src)
    tap = js.Module("amplitude", {}, {"signal": eng.fft.output("signal")}, "tap")  # reads the FFT
    rt = js.Runtime(eng.modules + [tap], fuse=True)
    assert not any(u.startswith("spectrum_fused") for u in rt.units)
    # the transform's output stays a tensor of its own (two readers); the engine's amplitude -> range pair, whose
    # intermediate nobody else reads, still runs as one pass (round 5: chain_fusions.cc)
    assert any(u.startswith("amplitude_range(") for u in rt.units), rt.units
    rt.compute()
    ref = oracle.spectrum_chain(x, -100.0, 0.0)
    assert_bit_equal(eng.fft.output("signal").numpy(), ref["fft"])
    assert_bit_equal(tap.output("signal").numpy(), ref["amplitude"])
    assert_bit_equal(eng.buffer.numpy(), ref["range"])


def test_graph_replay_equals_eager_and_timing_reports(js):
    x = csignal(np.random.default_rng(3), (64, 4096))
    outs = []
    for graph in (False, True):
        src = js.Tensor.from_numpy(x, sample=1, batch=0)
        eng = js.SpectrumEngine(src)
        spec = js.Module("spectrogram", {"height": 256}, {"signal": eng.buffer})
        rt = js.Runtime(eng.modules + [spec], graph=graph, fuse=True, timing=True)
        rt.compute(3)
        outs.append((eng.buffer.numpy(), spec.state("frequencyBins").numpy()))
        ms = rt.unit_mean_ms("spectrum_fused")
        assert 0.0 < ms < 50.0, ms
        if any(u.startswith("spectrogram") for u in rt.units):  # else it rides on the spectrum launches (one unit)
            assert rt.unit_mean_ms("spectrogram") > 0.0
        else:
            assert any(u.startswith("spectrum_fused_spectrogram(") for u in rt.units) and spec.timing["computeTime"] > 0.0
        assert eng.fft.timing["computeTime"] > 0.0
    assert_bit_equal(outs[0][0], outs[1][0])
    assert_bit_equal(outs[0][1], outs[1][1])


def test_direct_context_hooks(js, oracle):
    """computeInitialize / computeSubmit(stream) / computeDeinitialize called by a foreign harness
    (what TestContext / Benchmark do, src/testing.cc:123-140, src/benchmark.cc:145-160)."""
    import ctypes as C
    import torch

    x = csignal(np.random.default_rng(4), (2, 512))
    m = js.Module("fft", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    assert js._lib.jst_module_compute_initialize(m._h) == 0
    stream = torch.cuda.Stream()
    assert js._lib.jst_module_compute_submit(m._h, C.c_void_p(stream.cuda_stream)) == 0
    stream.synchronize()
    assert_bit_equal(m.output("signal").numpy(), oracle.fft_c2c(x))
    assert js._lib.jst_module_compute_deinitialize(m._h) == 0


def test_borrowed_torch_memory(js, oracle):
    import torch

    x = csignal(np.random.default_rng(5), (3, 2048))
    tx = torch.from_numpy(x).cuda()
    t = js.Tensor.wrap(tx.data_ptr(), tx.numel() * 8, "hip", "CF32", (3, 2048)).set_axes(sample=1, batch=0)
    m = js.Module("fft", {}, {"signal": t})
    rt = js.Runtime([m])
    rt.compute()
    assert_bit_equal(m.output("signal").numpy(), oracle.fft_c2c(x))


def test_timing_keeps_sampling_when_calls_are_shorter_than_a_period(js, oracle):
    """TIMING | GRAPH with a ring period > 1 and a caller that submits fewer cycles than a period per call: the
    eager timed cycle must still run at every sixteenth period boundary (it used to need a whole period inside
    one call, so Module::Timing froze on the cold settle sample while timing.cycles kept growing), the rest of
    the call replays as span graphs, and the results stay those of the oracle."""
    n, b, h, slots = 1024, 8, 64, 4
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "source")
    out = src.output("buffer")
    rng = np.random.default_rng(11)
    data = [csignal(rng, (b, n), 0.05) for _ in range(slots)]
    for s in range(slots):
        out.ring_select(s).copy_from(data[s])
    out.ring_select(0)
    eng = js.SpectrumEngine(out)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime([src] + eng.modules + [spec], graph=True, fuse=True, timing=True)
    assert rt.period == slots
    refs = [oracle.spectrum_chain(d, -100.0, 0.0)["range"] for d in data]
    bins = np.zeros(n * h, np.float32)
    seen, total = [], 0
    for chunk in [1] * 70 + [3] * 44 + [2] * 40:   # never a whole period in one call
        rt.compute(chunk)
        for _ in range(chunk):
            oracle.spectrogram(bins, refs[total % slots], h)
            total += 1
        if total in (1, 70, 202, 282):
            seen.append(eng.fft.timing["computeTime"])
    assert eng.fft.timing["cycles"] == total == 282
    assert 0.0 < seen[0] < seen[1] < seen[2] < seen[3], seen   # a new sample at least every 16 periods = 64 cycles
    assert_bit_equal(eng.buffer.numpy(), refs[(total - 1) % slots], "range output")
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, "spectrogram state")
    assert rt.graph_active
    rt.destroy()


def test_runtime_recreate_drops_span_graphs(js, oracle):
    """destroy() releases the span-graph cache with everything else: a second runtime over NEW modules that reaches
    the same (phase, length) keys must capture its own graphs, not replay stale ones."""
    n, b, slots = 512, 4, 4
    outs = []
    for seed in (1, 2):
        src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "source")
        buf = src.output("buffer")
        rng = np.random.default_rng(seed)
        data = [csignal(rng, (b, n), 0.05) for _ in range(slots)]
        for s in range(slots):
            buf.ring_select(s).copy_from(data[s])
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf)
        rt = js.Runtime([src] + eng.modules, graph=True, fuse=True)
        for chunk in (1, 2, 3, 2, 3, 1):   # spans at several (phase, length) keys
            rt.compute(chunk)
        outs.append(eng.buffer.numpy())
        assert_bit_equal(outs[-1], oracle.spectrum_chain(data[(12 - 1) % slots], -100.0, 0.0)["range"], f"runtime {seed}")
        rt.destroy()
        rt.destroy()   # idempotent
    assert not np.array_equal(outs[0], outs[1])
