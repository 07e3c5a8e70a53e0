"""The runtime's result convention (src/runtime/native/cpu/impl.cc:98-148) and the two modules that use it
here: SKIP (squelch, dsp/squelch/module_impl_native_cpu.cc:66-98) leaves everything downstream of the
skipping module out of the cycle while the other branches run; YIELD (a live source without data, like
io/soapy/module_impl_native_cpu.cc:47-60) ends the cycle quietly."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def test_squelch_gates_its_branch_only(js, oracle):
    rng = np.random.default_rng(21)
    n = 2048
    src = js.Tensor.create("hip", "CF32", (4, n)).set_axes(batch=0, sample=1)
    sq = js.Module("squelch", {"threshold": 0.5}, {"signal": src}, "squelch")
    fm = js.Module("fm", {"sampleRate": 240e3}, {"signal": sq.output("signal")}, "fm")      # gated branch
    amp = js.Module("amplitude", {}, {"signal": src}, "amplitude")                          # sibling of squelch
    rt = js.Runtime([sq, fm, amp], graph=True)
    assert not rt.graph_active
    loud = csignal(rng, (4, n), 1.0)
    quiet = csignal(rng, (4, n), 0.01)
    quiet[2, 7] = complex(np.nan, 0.3)   # NaN never becomes the peak
    history = []
    for x in (loud, quiet, quiet, loud):
        src.copy_from(x)
        assert rt.compute(1) == "success"
        passing, peak = oracle.squelch(x, 0.5)
        assert_bit_equal(sq.state("amplitude").numpy(), np.array([peak], np.float32), "peak amplitude")
        history.append(passing)
        assert_bit_equal(amp.output("signal").numpy(), oracle.amplitude(x, n), "sibling branch runs every cycle")
    assert history == [True, False, False, True]
    assert not rt.graph_active                                        # the decision is the host's: eager
    assert sq.timing["cycles"] == 4 and amp.timing["cycles"] == 4 and fm.timing["cycles"] == 2
    # the FM state saw exactly the two loud buffers, back to back
    ref = oracle.FmLane("narrow", "none", 240e3)
    ref(loud)
    assert np.max(np.abs(fm.output("signal").numpy().reshape(-1) - ref(loud).reshape(-1))) <= 2e-6
    # reference KATs (dsp/squelch/module_tests.cc): threshold 0 passes any non-zero buffer, blocks silence
    ones = js.Tensor.from_numpy(np.ones(16, np.float32))
    m = js.Module("squelch", {"threshold": 0.0}, {"signal": ones})
    f = js.Module("multiply_constant", {"constant": 3.0}, {"factor": m.output("signal")})
    r = js.Runtime([m, f])
    r.compute(1)
    assert f.timing["cycles"] == 1 and np.all(f.output("product").numpy() == 3.0)
    ones.copy_from(np.zeros(16, np.float32))
    r.compute(1)
    assert f.timing["cycles"] == 1                                     # 0 > 0 is false: skipped
    assert m.reconfigure({"threshold": 2.0}) == "success"
    with pytest.raises(js.JetstreamError, match="Invalid threshold"):
        js.Module("squelch", {"threshold": -1.0}, {"signal": ones})
    with pytest.raises(js.JetstreamError, match="Unsupported data type"):
        js.Module("squelch", {}, {"signal": js.Tensor.create("hip", "I32", (8,))})


def test_live_source_yields_without_data(js, oracle):
    rng = np.random.default_rng(22)
    n, b, slots = 1024, 4, 3
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots, "live": True}, {}, "sdr")
    eng = js.SpectrumEngine(src.output("buffer"))
    rt = js.Runtime([src] + eng.modules, graph=True, fuse=True)
    assert rt.compute(1) == "yield" and eng.fft.timing["cycles"] == 0     # nothing published yet
    out = src.output("buffer")
    data = [csignal(rng, (b, n), 0.1) for _ in range(5)]
    published = 0
    for k, x in enumerate(data):
        out.ring_select(published % slots).copy_from(x)
        published += 1
        assert src.reconfigure({"published": published}) == "success"
        assert rt.compute(1) == "success"
        assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -100.0, 0.0)["range"], f"buffer {k}")
        assert rt.compute(3) == "yield"                                    # one buffer, one cycle
    assert eng.fft.timing["cycles"] == 5 and not rt.graph_active
    assert src.reconfigure({"slots": 4}) == "recreate"
