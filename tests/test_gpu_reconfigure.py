"""Module::reconfigure (src/module.cc:233-290) on the hot-path modules: what the reference changes in place
(range min/max, multiply_constant constant, lineplot averaging, agc parameters, signal_generator waveform
parameters) is applied without a rebuild and reaches a captured hipGraph on the next compute(); what it
cannot change in place answers RECREATE and leaves the module untouched; an invalid configuration is
rejected with the staged one intact."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def test_range_reconfigures_in_place_through_a_captured_fused_graph(js, oracle):
    rng = np.random.default_rng(3)
    x = csignal(rng, (6, 4096), 0.05)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0)
    spec = js.Module("spectrogram", {"height": 64}, {"signal": eng.buffer})
    rt = js.Runtime(eng.modules + [spec], graph=True, fuse=True)
    rt.compute(3)
    assert rt.graph_active
    assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -100.0, 0.0)["range"], "before")
    assert eng.range.reconfigure({"min": -80.0, "max": -10.0}) == "success"
    assert eng.range.reconfigure({"min": -80.0, "max": -10.0}) == "success"      # unchanged: a no-op
    rt.compute(2)                                                                 # graph re-captured
    assert rt.graph_active
    assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -80.0, -10.0)["range"], "after")
    assert eng.range.reconfigure({"min": 5.0}, validate_only=True) == "success"   # nothing applied
    rt.compute(1)
    assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -80.0, -10.0)["range"], "validate only")
    with pytest.raises(js.JetstreamError, match="Invalid min/max"):
        eng.range.reconfigure({"min": "not-a-number"})
    rt.compute(1)
    assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -80.0, -10.0)["range"], "after a rejected one")
    # modules without an in-place path answer RECREATE like the reference's (amplitude, fft, spectrogram ...)
    assert eng.fft.reconfigure({"forward": False}) == "recreate"
    assert spec.reconfigure({"height": 128}) == "recreate"
    rt.compute(1)
    assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -80.0, -10.0)["range"], "after RECREATE answers")
    rt.destroy()


def test_in_place_and_recreate_answers_of_the_other_modules(js, oracle):
    rng = np.random.default_rng(4)
    x = csignal(rng, (3, 256))
    t = js.Tensor.from_numpy(x, batch=0, sample=1)
    mc = js.Module("multiply_constant", {"constant": 2.0}, {"factor": t})
    rt = js.Runtime([mc], graph=True)
    rt.compute(2)
    assert mc.reconfigure({"constant": 0.25}) == "success"
    rt.compute(2)
    ref = (x.real * np.float32(0.25) + 1j * (x.imag * np.float32(0.25))).astype(np.complex64)
    assert_bit_equal(mc.output("product").numpy(), ref)
    rt.destroy()
    f = rng.standard_normal((4, 512)).astype(np.float32)
    lp = js.Module("lineplot", {"averaging": 2}, {"signal": js.Tensor.from_numpy(f, batch=0, sample=1)})
    assert lp.reconfigure({"averaging": 8}) == "success"
    assert lp.reconfigure({"decimation": 2}) == "recreate"
    wf = js.Module("waterfall", {"height": 32}, {"signal": js.Tensor.from_numpy(f, batch=0, sample=1)})
    assert wf.reconfigure({"interpolate": False}) == "success"
    assert wf.reconfigure({"height": 64}) == "recreate"
    gen = js.Module("signal_generator", {"signalType": "cosine", "signalDataType": "CF32", "sampleRate": 1.0e6,
                                         "frequency": 1000.0, "bufferSize": 1024}, {})
    rt = js.Runtime([gen], graph=True)
    rt.compute(2)
    assert gen.reconfigure({"frequency": 12500.0, "amplitude": 0.5}) == "success"
    assert gen.reconfigure({"bufferSize": 2048}) == "recreate"
    assert gen.reconfigure({"signalType": "square"}) == "recreate"
    with pytest.raises(js.JetstreamError, match="Frequency"):
        gen.reconfigure({"frequency": 9.0e6})
    rt.compute(1)
    got = gen.output("signal").numpy()
    assert np.max(np.abs(got)) <= 0.5 + 1e-6 and np.max(np.abs(got)) > 0.45   # the new amplitude is live
    rt.destroy()
    agc = js.Module("agc", {"tileSize": 64}, {"signal": t})
    assert agc.reconfigure({"reference": 0.5, "maxGain": 10.0}) == "success"
    assert agc.reconfigure({"tileSize": 128}) == "recreate"
