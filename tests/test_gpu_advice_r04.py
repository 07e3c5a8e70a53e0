"""GPU: regressions for the round-3 advisor findings.
  * the live ring's slot-free event is recorded by the COMPUTE thread once the whole cycle is enqueued
    (Module::cycleSubmitted), never by the producer thread in front of kernels that were not submitted yet: a two-slot
    ring under a producer that runs ahead never mixes two batches in one cycle;
  * ringClear / a re-created runtime leave no stale "pending" slot or stream behind;
  * the real-operand (two products per sample) instantiation of the fused kernel is only chosen for an operand a
    statically settled unit produced: an operand that turns complex in a later cycle is multiplied in full;
  * a span-capable module ordered in front of the spectrum unit (a Lineplot of the static window) keeps the runtime
    per cycle instead of failing in compute();
  * ring_push refuses an array whose dtype is not the source's sample format."""
import threading
import time

import numpy as np
import pytest

from test_gpu_chain import tone_batch
from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def live_source(js, b, n, slots, **cfg):
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots, "live": True, **cfg}, {}, "sdr")
    return src, src.output("buffer")


def test_two_slot_ring_never_mixes_batches(js, oracle):
    """slots = 2 in steady state: the slot the producer wants next is always the one the latest cycle consumed."""
    n, b, slots, total = 4096, 32, 2, 60
    src, out = live_source(js, b, n, slots, overflow="reject")
    amp = js.Module("amplitude", {}, {"signal": out}, "amp")
    rge = js.Module("range", {"min": -100.0, "max": 100.0}, {"signal": amp.output("signal")}, "range")
    water = js.Module("waterfall", {"height": total * b}, {"signal": rge.output("signal")}, "history")
    rt = js.Runtime([src, amp, rge, water])
    payload = [np.full((b, n), k + 1, np.complex64) for k in range(total)]

    def producer():
        for x in payload:
            while src.ring_push(x) == "incomplete":
                pass                                    # spin: be at the slot the moment it is consumed
    th = threading.Thread(target=producer)
    th.start()
    cycles, deadline = 0, time.time() + 120
    while cycles < total and time.time() < deadline:
        if src.ring_wait(b * n, timeout_ms=200):
            assert rt.compute(1, sync=False) == "success"
            cycles += 1
    th.join()
    rt.synchronize()
    assert cycles == total
    rows = water.state("frequencyBins").numpy().reshape(total, b, n)
    for k in range(total):
        want = oracle.range_(oracle.amplitude(np.full((1, 1), k + 1, np.complex64), n), -100.0, 100.0)[0, 0]
        assert np.all(rows[k] == want), f"batch {k}: {np.unique(rows[k])[:4]} instead of {want}"
    rt.destroy()


def test_ring_clear_and_runtime_recreation_leave_no_stale_state(js, oracle):
    n, b, slots = 1024, 8, 2
    src, out = live_source(js, b, n, slots)
    amp = js.Module("amplitude", {}, {"signal": out}, "amp")
    x = [np.full((b, n), 3 + k, np.complex64) for k in range(6)]
    rt = js.Runtime([src, amp])
    assert src.ring_push(x[0]) == "success"
    rt.compute(1)
    rt.destroy()                                        # the stream the source remembered is gone
    assert src.ring_push(x[1]) == "success" and src.ring_push(x[2]) == "success"   # both slots: no event on a dead stream
    rt = js.Runtime([src, amp])
    rt.compute(1)
    assert_bit_equal(amp.output("signal").numpy(), oracle.amplitude(x[1], n), "first batch after the runtime came back")
    src.ring_clear()
    assert src.ring_size == 0
    assert src.ring_push(x[3]) == "success" and src.ring_push(x[4]) == "success"
    rt.compute(1)
    assert_bit_equal(amp.output("signal").numpy(), oracle.amplitude(x[3], n), "first batch after ring_clear")
    rt.compute(1)
    assert_bit_equal(amp.output("signal").numpy(), oracle.amplitude(x[4], n), "second batch after ring_clear")
    rt.destroy()


def test_operand_that_turns_complex_is_multiplied_in_full(js, oracle):
    """The Multiply operand comes from a tensor NO module produces (not statically settled): real in cycle 1, complex
    in cycle 2.  Provider fast must not have latched the real-operand kernel."""
    n, b, h = 4096, 16, 256
    x = tone_batch(oracle, b, n, 5) * np.float32(0.3)
    w_real = np.ascontiguousarray(oracle.invert(oracle.window(n)).reshape(1, n))
    assert not np.any(w_real.imag)
    rng = np.random.default_rng(3)
    w_cplx = (w_real + 1j * (0.4 * rng.standard_normal((1, n))).astype(np.float32)).astype(np.complex64)
    sig = js.Tensor.from_numpy(x, sample=1, batch=0)
    win = js.Tensor.from_numpy(w_real, sample=1)
    mul = js.Module("multiply", {}, {"a": sig, "b": win}, "multiply")
    fft = js.Module("fft", {"forward": True}, {"signal": mul.output("product")}, "fft")
    amp = js.Module("amplitude", {}, {"signal": fft.output("signal")}, "amplitude", provider="fast")
    rge = js.Module("range", {"min": -100.0, "max": 0.0}, {"signal": amp.output("signal")}, "range", provider="fast")
    spec = js.Module("spectrogram", {"height": h}, {"signal": rge.output("signal")}, "spectrogram")
    rt = js.Runtime([mul, fft, amp, rge, spec], fuse=True, graph=False)
    assert any(u.startswith("spectrum_fused(") for u in rt.units), rt.units
    for w in (w_real, w_cplx, w_real):
        win.copy_from(w)
        rt.compute(1)
        ref = oracle.range_(oracle.amplitude(oracle.fft_c2c(oracle.multiply(x, w)), n), -100.0, 0.0)
        assert np.max(np.abs(rge.output("signal").numpy() - ref)) <= 4e-7
    rt.destroy()


def test_span_capable_module_in_front_of_the_spectrum_unit_keeps_the_runtime_per_cycle(js, oracle):
    n, b, h, slots = 4096, 8, 64, 3
    xs = [tone_batch(oracle, b, n, 7 + s) * np.float32(0.2 + 0.2 * s) for s in range(slots)]
    ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "ring")
    buf = ring.output("buffer")
    for s, x in enumerate(xs):
        buf.ring_select(s).copy_from(x)
    buf.ring_select(0)
    # a Lineplot whose input is NOT the spectrum unit's output, ordered first: Lineplot::spanCapable() is true
    other = js.Tensor.from_numpy(np.abs(xs[0]).astype(np.float32), sample=1, batch=0)
    plot = js.Module("lineplot", {"averaging": 1}, {"signal": other}, "plot_of_something_else")
    eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime([ring, plot] + eng.modules + [spec], fuse=True, graph=True, batch=True)
    assert not rt.batched
    rt.compute(1)
    rt.compute(2 * slots)                               # works per cycle (it used to fail inside the batched submission)
    refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in xs]
    bins = np.zeros(n * h, np.float32)
    for k in range(1 + 2 * slots):
        oracle.spectrogram(bins, refs[k % slots], h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(-1), bins, "per-cycle runtime, state after 7 cycles")
    rt.destroy()


def test_ring_push_checks_the_sample_format(js):
    src, _ = live_source(js, 4, 256, 2)
    with pytest.raises(js.JetstreamError, match="takes"):
        src.ring_push(np.zeros((4, 256, 2), np.int8))     # CF32 source: 4x too few bytes behind this pointer
    with pytest.raises(js.JetstreamError, match="takes"):
        src.ring_push(np.zeros((4, 256, 2), np.float32))  # would be counted as twice the elements
    src16, _ = live_source(js, 4, 256, 2, dtype="CI16")
    with pytest.raises(js.JetstreamError, match="takes"):
        src16.ring_push(np.zeros((4, 256), np.complex64))
    assert src16.ring_push(np.zeros((4, 256, 2), np.int16)) == "success"


def test_single_rank_communicator_on_device_tensors(js):
    """jst_comm_allreduce with world = 1: counted, a no-op on the data (sum and average alike), operand rules enforced."""
    c = js.Comm(0, 1, None)
    counts = js.Tensor.from_numpy(np.arange(64 * 128, dtype=np.uint32).reshape(64, 128))
    trace = js.Tensor.from_numpy(np.linspace(-3, 3, 4096, dtype=np.float32))
    c.all_reduce(counts, "sum")
    c.all_reduce(trace, "sum", average=True)
    c.all_reduce(trace, "max")
    assert c.calls == 3 and not c.uses_rccl
    assert np.array_equal(counts.numpy(), np.arange(64 * 128, dtype=np.uint32).reshape(64, 128))
    assert np.array_equal(trace.numpy(), np.linspace(-3, 3, 4096, dtype=np.float32))
    with pytest.raises(js.JetstreamError, match="dense F32 or U32 HIP tensor"):
        c.all_reduce(js.Tensor.from_numpy(np.zeros((4, 8), np.complex64)))
    with pytest.raises(js.JetstreamError, match="average"):
        c.all_reduce(counts, "sum", average=True)
    sliced = js.Tensor.from_numpy(np.zeros((8, 8), np.float32)).slice(1, 0, 4)
    with pytest.raises(js.JetstreamError, match="dense F32 or U32 HIP tensor"):
        c.all_reduce(sliced)
