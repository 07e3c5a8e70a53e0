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
    rt.compute()
    assert_bit_equal(tap.output("signal").numpy(), eng.amplitude.output("signal").numpy())
    assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -100.0, 0.0)["range"])


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
