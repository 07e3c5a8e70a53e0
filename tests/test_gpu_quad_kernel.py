"""fft_quad_kernel (cyberether_amd/csrc/kernels/fft_quad.hh): the round-5 4096-point side kernel of provider fast -- 256 threads
per transform, in-place exchange, rows by LDS-DMA, four workgroups per CU, the last rounds of a long launch handed out from
device counters.  It must leave exactly what fft_pipe_kernel leaves (same pocketfft arithmetic, another schedule): every F32
value, every row-index byte, every word of the Spectrogram state -- per cycle, cycle-batched with the static round robin
(short launches) and with claimed rounds (launches of eight rounds of 4 x CUs transforms and more), and against the oracle
(bins exact, floats within provider fast's tolerance, fft/module_impl_native_cpu.cc:125-140 + amplitude + range +
spectrogram/module_impl_native_cpu.cc:61-87)."""
import numpy as np
import pytest

from test_gpu_batch import _ring_chain
from test_gpu_chain import tone_batch
from test_gpu_fast_provider import RANGE_TOL_ABS
from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def _run(js, xs, h, calls, switch, kernel="quad", static=False, batch=True):
    switch("JST_FFT_KERNEL", kernel or None)
    switch("JST_QUAD_STATIC", "1" if static else None)
    eng, spec, rt = _ring_chain(js, xs, h, provider="fast", batch=batch)
    assert rt.batched == batch
    assert any(u.startswith("spectrum_fused(") and "+indices" in u for u in rt.units), rt.units
    trace, done = [], 0
    for call in calls:
        rt.compute(call)
        done += call
        latest = eng.buffer.numpy().copy()  # the handle shows the last cycle's output
        slots = None
        if batch and call >= len(xs):  # every slot of the output ring was written by this call
            slots = [eng.buffer.ring_select(s).numpy().copy() for s in range(len(xs))]
            eng.buffer.ring_select((done - 1) % len(xs))
        trace.append((latest, spec.state("frequencyBins").numpy().copy(), slots))
    rt.destroy()
    return trace


def _same(a, b, what):
    for i, ((o0, s0, r0), (o1, s1, r1)) in enumerate(zip(a, b)):
        assert_bit_equal(o1, o0, f"{what}: output after call {i}")
        assert_bit_equal(s1, s0, f"{what}: spectrogram state after call {i}")
        if r0 is not None and r1 is not None:
            for s, (x0, x1) in enumerate(zip(r0, r1)):
                assert_bit_equal(x1, x0, f"{what}: output ring slot {s} after call {i}")


@pytest.mark.parametrize("b,slots,h", [(64, 4, 256), (3, 5, 100), (1030, 2, 255)])
def test_quad_equals_pipe_bit_for_bit(js, oracle, switch, b, slots, h):
    """Short launches (static round robin): per cycle and cycle-batched; ragged batch counts (3: fewer transforms than
    workgroups; 1030: one more round for six workgroups)."""
    n = 4096
    xs = [tone_batch(oracle, b, n, 21 + s) * np.float32(0.2 + 0.3 * s) for s in range(slots)]
    xs[0][1, :] = 0  # a row of zeros and a non-finite sample: the guard's exact ladder / NaN propagation
    xs[-1][2 % b, 99] = np.complex64(complex(np.nan, 1.0))
    calls = (1, slots, 2, 2 * slots + 1, 3)
    pipe = _run(js, xs, h, calls, switch, kernel="pipe")
    quad = _run(js, xs, h, calls, switch)
    _same(pipe, quad, "quad vs pipe, cycle-batched")
    quad_pc = _run(js, xs, h, calls, switch, batch=False)
    for i, ((o0, s0, _), (o1, s1, _)) in enumerate(zip(pipe, quad_pc)):
        assert_bit_equal(o1, o0, f"quad per cycle vs pipe: output after call {i}")
        assert_bit_equal(s1, s0, f"quad per cycle vs pipe: state after call {i}")


def test_quad_claimed_rounds_equal_static_and_pipe(js, oracle, switch):
    """A launch long enough for the claimed rounds (the bench's own shape: 1024 x 4096 per cycle, spans of 8+ cycles on a
    256-CU device): the device counters hand every transform out exactly once, and re-arm themselves for the next launch."""
    n, b, slots, h = 4096, 1024, 12, 256
    base = tone_batch(oracle, b, n, 77)
    xs = [np.roll(base, 37 * s, axis=0) * np.float32(0.5 + 0.05 * s) for s in range(slots)]
    calls = (1, slots, slots, 2 * slots + 5, 9)
    dyn = _run(js, xs, h, calls, switch)
    sta = _run(js, xs, h, calls, switch, static=True)
    _same(sta, dyn, "claimed rounds vs static round robin")
    pipe = _run(js, xs, h, calls, switch, kernel="pipe")
    _same(pipe, dyn, "claimed rounds vs fft_pipe_kernel")
    # the library's own choice (no JST_FFT_KERNEL): fft_pipe_kernel for the short launches, the quad kernel from eight rounds
    auto = _run(js, xs, h, calls, switch, kernel=None)
    _same(pipe, auto, "the default selection by launch size")
    # ... and the bins are the reference's (provider generic = every float of the CPU path), the floats within tolerance
    ref = oracle.spectrum_chain(xs[(sum(calls) - 1) % slots], -100.0, 0.0)["range"]
    got = dyn[-1][0]
    assert np.max(np.abs(got - ref)) <= RANGE_TOL_ABS
    assert np.array_equal((got * np.float32(h)).astype(np.uint64), (ref * np.float32(h)).astype(np.uint64))
