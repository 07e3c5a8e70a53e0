"""The producer side of a live ring_source against the guarantees of the reference's CircularBuffer
(include/jetstream/tools/circular_buffer.hh:31-48, src/tools/circular_buffer.cc) as the Soapy module uses it
(<= 8192-sample pushes, soapy/module_impl.cc:375-399; waitForSize + pop of one batch per cycle,
module_impl_native_cpu.cc:39-60): arbitrary chunk sizes, wrap of the slot ring, partial batches, both overflow
policies, YIELD without data, the blocking wait, the zero-copy acquire / commit form, integer SDR sample formats,
and -- the guarantee the caller used to provide by hand -- no upload lands in a slot a cycle is still reading."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def make(js, b, n, slots, **cfg):
    src = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots, "live": True, **cfg}, {}, "sdr")
    return src, src.output("buffer")


def test_chunks_of_any_size_wrap_partial_and_yield(js, oracle):
    rng = np.random.default_rng(5)
    n, b, slots = 1024, 4, 3
    src, out = make(js, b, n, slots)
    eng = js.SpectrumEngine(out)
    rt = js.Runtime([src] + eng.modules, graph=True, fuse=True)
    assert rt.compute(1) == "yield" and src.ring_size == 0 and src.ring_capacity == slots * b * n
    stream = csignal(rng, (7 * b * n + 1234,), 0.1)        # seven batches and a partial one
    import itertools
    sizes = itertools.cycle((8192, 1, 4095, 8192, 3 * b * n, 8192, 777))   # Soapy-sized, tiny, odd, larger than a batch
    pos, cycles = 0, 0
    while pos < stream.size:
        take = min(next(sizes), stream.size - pos)
        assert src.ring_push(stream[pos:pos + take]) == "success"
        pos += take
        while src.ring_size >= b * n:                       # consume every complete batch (the slot ring wraps)
            assert rt.compute(1) == "success"
            ref = oracle.spectrum_chain(stream[cycles * b * n:(cycles + 1) * b * n].reshape(b, n), -100.0, 0.0)["range"]
            assert_bit_equal(eng.buffer.numpy(), ref, f"batch {cycles}")
            cycles += 1
        assert rt.compute(1) == "yield"                     # less than a batch buffered: no cycle
    assert cycles == 7 and src.ring_size == 1234 and src.ring_overflows == 0
    assert eng.fft.timing["cycles"] == 7
    rt.destroy()


def test_overflow_overwrite_drops_the_oldest_and_reject_refuses(js):
    n, b, slots = 256, 2, 2
    batches = [np.full((b, n), k + 1, np.complex64) for k in range(6)]
    src, out = make(js, b, n, slots)                        # default policy: overwrite (OverwriteOldest)
    rt = js.Runtime([src])
    for x in batches[:5]:
        assert src.ring_push(x) == "success"
    assert src.ring_overflows == 3 and src.ring_size == slots * b * n   # batches 0..2 were dropped
    for want in (4, 5):
        assert rt.compute(1) == "success"
        assert np.all(out.numpy() == want)
    assert rt.compute(1) == "yield"
    src.ring_clear()
    assert src.ring_overflows == 0 and src.ring_size == 0

    src2, out2 = make(js, b, n, slots, overflow="reject")
    rt2 = js.Runtime([src2])
    assert src2.ring_push(batches[0]) == "success" and src2.ring_push(batches[1]) == "success"
    assert src2.ring_push(batches[2]) == "incomplete"       # would complete a third batch: nothing taken
    assert src2.ring_push(batches[2][:1, :100]) == "success"   # a partial batch still fits the staging memory
    assert src2.ring_overflows == 1 and src2.ring_size == 2 * b * n + 100
    assert rt2.compute(1) == "success" and np.all(out2.numpy() == 1)
    rest = np.concatenate([batches[2].reshape(-1)[100:]])
    assert src2.ring_push(rest) == "success"                # now there is a free slot again
    for want in (2, 3):
        assert rt2.compute(1) == "success" and np.all(out2.numpy() == want)
    with pytest.raises(js.JetstreamError):
        js.Module("ring_source", {"batches": b, "samples": n, "slots": 2, "live": True, "overflow": "drop"}, {})
    with pytest.raises(js.JetstreamError):
        js.Module("ring_source", {"batches": b, "samples": n, "slots": 2}, {}).ring_push(batches[0])   # not live


def test_wait_for_size_and_zero_copy_commit(js):
    n, b, slots = 512, 2, 4
    src, out = make(js, b, n, slots)
    rt = js.Runtime([src])
    t0 = time.perf_counter()
    assert src.ring_wait(b * n, timeout_ms=50) is False and time.perf_counter() - t0 >= 0.04
    x = csignal(np.random.default_rng(1), (b, n), 1.0)

    def producer():
        time.sleep(0.05)
        addr, room = src.ring_acquire()                     # the driver writes into the pinned staging memory itself
        assert room == b * n
        C.memmove(addr, x.ctypes.data, x.nbytes // 2)
        assert src.ring_commit(b * n // 2) == "success"
        addr2, room2 = src.ring_acquire()
        assert room2 == b * n // 2 and addr2 == addr + x.nbytes // 2
        C.memmove(addr2, x.ctypes.data + x.nbytes // 2, x.nbytes // 2)
        assert src.ring_commit(b * n // 2) == "success"
    th = threading.Thread(target=producer)
    th.start()
    assert src.ring_wait(b * n, timeout_ms=5000) is True    # soapy/module_impl_native_cpu.cc:39-45
    th.join()
    assert rt.compute(1) == "success"
    assert_bit_equal(out.numpy(), x, "zero-copy batch")
    with pytest.raises(js.JetstreamError):
        src.ring_commit(b * n + 1)


@pytest.mark.parametrize("kernel", ["pipe", "slot"])
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("dtype,np_t,scale", [("CI16", np.int16, 32768.0), ("CI8", np.int8, 128.0), ("CU8", np.uint8, 128.0)])
def test_integer_sample_formats_through_the_ring(js, oracle, switch, dtype, np_t, scale, fuse, kernel):
    """Raw SDR formats: the ring holds CI16 / CI8 / CU8 samples (4 or 2 bytes over PCIe instead of 8) and a cast module
    (the spectrum_engine block's own first module) turns them into CF32 (cast/module_impl.cc:49-70: divide by 32768 / 128).  Fused, the
    conversion happens in the transform's first load: the unit reads the cast's INPUT, the cast launches nothing, and
    the result is bit-identical to the module-by-module path and to the oracle."""
    switch("JST_FFT_KERNEL", kernel)
    n, b, slots = 1024, 4, 3
    rng = np.random.default_rng(9)
    info = np.iinfo(np_t)
    src, out = make(js, b, n, slots, dtype=dtype)
    assert out.dtype == dtype
    eng = js.SpectrumEngine(out)        # the block's own cast (spectrum_engine/block_impl.cc:120-217) takes the raw samples
    cast = eng.cast
    rt = js.Runtime([src] + eng.modules, graph=True, fuse=fuse)
    fused_names = [u for u in rt.units if u.startswith("spectrum_fused(")]
    assert (fused_names and fused_names[0].startswith("spectrum_fused(spectrum.cast_input+")) if fuse else not fused_names, rt.units
    for k in range(5):
        raw = rng.integers(info.min, info.max + 1, (b, n, 2)).astype(np_t)
        if k == 0:
            raw[0, :8] = [[info.min, info.max], [info.max, info.min], [0, 0], [info.min, info.min], [info.max, info.max],
                          [1, 0], [0, 1], [info.min, 0]]
        for piece in np.array_split(raw.reshape(-1, 2), 5):
            assert src.ring_push(piece) == "success"
        assert rt.compute(1) == "success"
        x = (raw[..., 0].astype(np.float32) / np.float32(scale) + 1j * (raw[..., 1].astype(np.float32) / np.float32(scale))).astype(np.complex64)
        if not fuse:
            assert_bit_equal(cast.output("buffer").numpy(), x, f"cast {dtype} batch {k}")
        assert_bit_equal(eng.buffer.numpy(), oracle.spectrum_chain(x, -100.0, 0.0)["range"], f"{dtype} batch {k}")
    assert cast.timing["cycles"] == 5
    rt.destroy()


def test_no_upload_lands_in_a_slot_its_cycle_is_still_reading(js, oracle):
    """A producer thread pushes as fast as it can into a 3-slot ring (policy reject: it retries, nothing is dropped)
    while the consumer submits cycle after cycle WITHOUT ever synchronising with the device (bench.py used to
    synchronise by hand every slots/2 cycles).  Batch k is the constant k + 1; the chain is amplitude -> waterfall
    with a waterfall tall enough to keep every batch, so the whole history is read back once at the end: rows of
    batch k must all carry amplitude(k + 1) -- an upload that overtook its slot's reader, or a cycle that ran ahead
    of its upload, would leave a mixed, repeated or stale block."""
    n, b, slots, total = 4096, 32, 3, 40
    src, out = make(js, b, n, slots, overflow="reject")
    amp = js.Module("amplitude", {}, {"signal": out}, "amp")
    water = js.Module("waterfall", {"height": total * b}, {"signal": amp.output("signal")}, "history")
    rt = js.Runtime([src, amp, water])

    payload = [np.full((b, n), k + 1, np.complex64) for k in range(total)]

    def producer():
        for x in payload:
            while src.ring_push(x) == "incomplete":   # ring full: retry (nothing is dropped under "reject")
                time.sleep(0.0001)
    th = threading.Thread(target=producer)
    th.start()
    cycles, deadline = 0, time.time() + 120
    while cycles < total and time.time() < deadline:
        if cycles in (5, 20):
            time.sleep(0.02)                            # a slow consumer for a moment: the producer fills the ring
        if src.ring_wait(b * n, timeout_ms=200):
            assert rt.compute(1, sync=False) == "success"   # never a host synchronise between cycles and uploads
            cycles += 1
    th.join()
    rt.synchronize()
    assert cycles == total
    rows = water.state("frequencyBins").numpy().reshape(total, b, n)
    for k in range(total):
        want = oracle.amplitude(np.full((1, 1), k + 1, np.complex64), n)[0, 0]
        assert np.all(rows[k] == want), f"batch {k}: {np.unique(rows[k])[:4]} instead of {want}"
    assert src.ring_overflows > 0          # the producer did run ahead: the ring was full at times
    rt.destroy()


def test_a_cycle_that_ends_early_still_closes_its_slot(js):
    """a cycle that ends in YIELD behind the source's computeSubmit (here: a SECOND live source without
    data) used to leave the first source's slot waiting for a completion nobody recorded -- the producer's next push to
    that slot sat out the 200 ms timeout.  The runtime now closes the cycle on its early returns (jst/module.cc submitAll)."""
    n, b = 1024, 2
    a = js.Module("ring_source", {"batches": b, "samples": n, "slots": 1, "live": True}, {}, "a")
    bsrc = js.Module("ring_source", {"batches": b, "samples": n, "slots": 1, "live": True}, {}, "b")
    amp_a = js.Module("amplitude", {}, {"signal": a.output("buffer")}, "amp_a")
    amp_b = js.Module("amplitude", {}, {"signal": bsrc.output("buffer")}, "amp_b")
    rt = js.Runtime([a, bsrc, amp_a, amp_b])
    x = np.ones((b, n), np.complex64)
    assert a.ring_push(x) == "success"
    assert rt.compute(1) == "yield"       # `a` took its slot, `b` has nothing: the cycle ends there
    rt.synchronize()
    t0 = time.perf_counter()
    assert a.ring_push(x) == "success"    # the only slot again
    assert time.perf_counter() - t0 < 0.1, "the push waited for a cycle that had already ended"
    rt.destroy()


def test_clear_while_a_push_waits_for_its_slot(js):
    """publishStagedBatch releases the ring's mutex while it waits for the slot's cycle to close; a
    ringClear() in that window resets what it computed before.  The staged batch must go with the rest of the ring
    (nothing uploaded, nothing counted), and the ring must keep working afterwards."""
    n, b = 1024, 2
    src, out = make(js, b, n, 1)
    amp = js.Module("amplitude", {}, {"signal": out}, "amp")
    rt = js.Runtime([src, amp])
    x = np.full((b, n), 3, np.complex64)
    assert src.ring_push(x) == "success"
    assert src.ring_size == b * n
    src.ring_clear()
    assert src.ring_size == 0 and rt.compute(1) == "yield"
    y = np.full((b, n), 5, np.complex64)
    assert src.ring_push(y) == "success" and rt.compute(1) == "success"
    got = amp.output("signal").numpy()
    assert np.all(got == got[0, 0]) and got[0, 0] != 0
    rt.destroy()
