"""JST_RUNTIME_COMBINE on the GPU: a Spectrogram that is the only reader of the fused spectrum unit's output rides on
the NEXT cycle's spectrum launch (one kernel per cycle, the output a ring of two slots, the spectrogram still waiting
when a compute call ends is run then).  Results and what is visible after every compute call must be exactly those of
the plain runtime: range output and spectrogram state vs the oracle, bit for bit, over eager cycles, period graphs
and span graphs, for chunk sizes that start and end anywhere in the ring."""
import numpy as np
import pytest

from test_gpu_chain import tone_batch
from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def chain(js, out, h, graph, combine, provider="generic", extra=None):
    eng = js.SpectrumEngine(out, provider=provider)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    mods = eng.modules + [spec]
    if extra == "lineplot":
        mods.append(js.Module("lineplot", {}, {"signal": eng.buffer}, "lineplot"))
    return eng, spec, mods


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("h", [256, 512, 64])
def test_combined_equals_oracle_over_chunks(js, oracle, graph, h):
    n, b, slots = 4096, 8, 4
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "source")
    out = src.output("buffer")
    data = [tone_batch(oracle, b, n, 900 + s) for s in range(slots)]
    for s in range(slots):
        out.ring_select(s).copy_from(data[s])
    out.ring_select(0)
    eng, spec, mods = chain(js, out, h, graph, True)
    rt = js.Runtime([src] + mods, graph=graph, fuse=True, combine=True)
    assert any(u.startswith("spectrum_fused_spectrogram(") for u in rt.units), rt.units
    assert not any(u.startswith("spectrogram") for u in rt.units)
    assert rt.period == slots  # lcm(ring slots, the two output slots)
    refs = [oracle.spectrum_chain(d, -100.0, 0.0)["range"] for d in data]
    bins = np.zeros(n * h, np.float32)
    total = 0
    for chunk in (1, 1, 2, 3, 4, 8, 5, 7, 1, 16, 9, 2, 4, 4, 13):
        rt.compute(chunk)
        for _ in range(chunk):
            oracle.spectrogram(bins, refs[total % slots], h)
            total += 1
        assert_bit_equal(eng.buffer.numpy(), refs[(total - 1) % slots], f"range output after {total} cycles")
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, f"bins after {total} cycles")
    assert spec.timing["cycles"] == total and eng.fft.timing["cycles"] == total
    if graph:
        assert rt.graph_active


def test_unsynchronised_calls_then_synchronize(js, oracle):
    n, b, h = 4096, 16, 256
    x = tone_batch(oracle, b, n, 31)
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng, spec, mods = chain(js, src, h, True, True)
    rt = js.Runtime(mods, graph=True, fuse=True, combine=True)
    assert rt.period == 2
    ref = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    bins = np.zeros(n * h, np.float32)
    for chunk in (3, 2, 6):
        rt.compute(chunk, sync=False)
        for _ in range(chunk):
            oracle.spectrogram(bins, ref, h)
    rt.synchronize()
    assert_bit_equal(eng.buffer.numpy(), ref)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins)


def test_fast_provider_bins_match_the_plain_runtime(js, oracle):
    n, b, h = 4096, 32, 256
    x = tone_batch(oracle, b, n, 77)
    states = []
    for combine in (False, True):
        src = js.Tensor.from_numpy(x, sample=1, batch=0)
        eng, spec, mods = chain(js, src, h, True, combine, provider="fast")
        rt = js.Runtime(mods, graph=True, fuse=True, combine=combine)
        assert any(u.startswith("spectrum_fused_spectrogram(") for u in rt.units) == combine
        rt.compute(5)
        states.append((eng.buffer.numpy(), spec.state("frequencyBins").numpy()))
    assert_bit_equal(states[0][0], states[1][0], "fast range output")
    assert_bit_equal(states[0][1], states[1][1], "fast spectrogram state")


@pytest.mark.parametrize("case", ["second_reader", "other_length", "flag_off"])
def test_falls_back_to_two_kernels(js, oracle, case):
    n = 2048 if case == "other_length" else 4096
    x = tone_batch(oracle, 4, n, 5)
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng, spec, mods = chain(js, src, 128, True, True, extra="lineplot" if case == "second_reader" else None)
    rt = js.Runtime(mods, graph=True, fuse=True, combine=case != "flag_off")
    assert not any(u.startswith("spectrum_fused_spectrogram(") for u in rt.units), rt.units
    assert any(u.startswith("spectrum_fused(") for u in rt.units) and any(u.startswith("spectrogram") for u in rt.units)
    rt.compute(3)
    ref = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    bins = np.zeros(n * 128, np.float32)
    for _ in range(3):
        oracle.spectrogram(bins, ref, 128)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins)
