"""Round 6: the LDS-tiled blocks kernel's PERSISTENT form (fft_tiled.hip: static plans, long launches -- a workgroup loops over
tiles with the next tile's elements prefetched into registers, the block passes' twiddles resident in LDS, plain stores) and
provider fast on the tiled path (config 5's plan as a constant for the lean amplitude / range epilogue).

Same arithmetic as one workgroup per tile, so: every output bit of the bit-exact provider against the oracle
(pocketfft order restated: pocketfft.hh:1476-1497, amplitude / range module_impl_native_cpu.cc) on launches long enough to take
the persistent kernel, the short launch of the same data bit for bit, and provider fast within north_star's 1e-5 (measured bound
4e-7) -- identical bits whichever of the two kernels ran."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu

N = 65536


def _signal(b, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(N)
    x = np.exp(2j * np.pi * (1000.25 + 37.5 * np.arange(b))[:, None] * t[None, :] / N)
    x = x + 1e-2 * (rng.standard_normal((b, N)) + 1j * rng.standard_normal((b, N)))
    return x.astype(np.complex64)


def _run(js, x, provider, slots=0, cycles=1):
    """slots == 0: a plain tensor, one launch per cycle; else a resident ring of `slots` rotations of x, cycle-batched."""
    b = x.shape[0]
    if slots:
        ring = js.Module("ring_source", {"batches": b, "samples": N, "slots": slots}, {}, "iq")
        buf = ring.output("buffer")
        for s in range(slots):
            buf.ring_select(s).copy_from(np.roll(x, s, axis=0))
        buf.ring_select(0)
        mods, src = [ring], buf
    else:
        mods, src = [], js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
    lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
    rt = js.Runtime(mods + eng.modules + [lp], graph=True, fuse=True, batch=bool(slots))
    rt._keep = mods + eng.modules + [lp]
    rt.compute(cycles)
    return eng, lp, rt


def test_long_launch_takes_the_persistent_kernel_bit_exact(js, oracle):
    """128 transforms x 4 ring slots = 8192 block tiles per span launch (>= 4 rounds of the chip's slots): the persistent kernel.
    Every slot of the output ring against the oracle, and against the one-launch-per-cycle runtime (one workgroup per tile)."""
    b, slots = 128, 4
    x = _signal(b, 5)
    ref = oracle.spectrum_chain(x[:8], -100.0, 0.0)["range"]
    ref_tail = oracle.spectrum_chain(x[-4:], -100.0, 0.0)["range"]
    eng, lp, rt = _run(js, x, "generic", slots=slots, cycles=1 + 2 * slots)
    assert rt.batched, rt.units
    for s in range(slots):
        got = np.roll(eng.buffer.ring_select(s).numpy(), -s, axis=0)  # slot s holds the spectra of np.roll(x, s)
        assert_bit_equal(got[:8], ref, f"slot {s}, rows 0-7 against the oracle")
        assert_bit_equal(got[-4:], ref_tail, f"slot {s}, last rows against the oracle")
        if s == 0:
            whole = got.copy()
        else:
            assert_bit_equal(got, whole, f"slot {s} equals slot 0 up to the rotation")
    rt.destroy()
    eng1, lp1, rt1 = _run(js, x, "generic")
    assert_bit_equal(eng1.buffer.numpy(), whole, "one workgroup per tile against the persistent kernel, all 128 x 65536 outputs")
    rt1.destroy()


@pytest.mark.parametrize("b,slots", [(16, 16), (128, 4)])
def test_provider_fast_on_the_tiled_path(js, oracle, b, slots):
    """Provider fast at 65536 points (config 5's static plan with the lean epilogue): within 1e-5 of the oracle (the bit-exact
    chain), and the cycle-batched span launches (persistent kernel) leave the very bits of the per-cycle launches."""
    x = _signal(b, 9)
    rows = np.r_[0:3, b - 3:b]
    ref = oracle.spectrum_chain(x[rows], -100.0, 0.0)["range"]
    eng1, lp1, rt1 = _run(js, x, "fast")
    per_cycle = eng1.buffer.numpy().copy()
    err = float(np.max(np.abs(per_cycle[rows] - ref)))
    assert err <= 1e-5, err
    assert err <= 4e-7, f"provider fast drifted: {err}"
    rt1.destroy()
    eng, lp, rt = _run(js, x, "fast", slots=slots, cycles=1 + slots)
    assert rt.batched, rt.units
    for s in (0, 1, slots - 1):
        got = np.roll(eng.buffer.ring_select(s).numpy(), -s, axis=0)
        assert_bit_equal(got, per_cycle, f"slot {s}: span launch against the per-cycle launch")
    rt.destroy()


def test_lineplot_behind_the_persistent_kernel(js, oracle):
    """The Lineplot's average over a long cycle-batched span equals the per-cycle runtime's, bit for bit (the sink reads what the
    persistent kernel stored plainly: visibility across the launch boundary)."""
    b, slots = 128, 4
    x = _signal(b, 21)
    states = []
    for ring_slots in (0, slots):
        if ring_slots:
            eng, lp, rt = _run(js, x, "generic", slots=ring_slots, cycles=1 + 2 * ring_slots)
        else:
            # the same sequence of inputs per cycle: rotations of x
            eng, lp, rt = None, None, None
            ring = js.Module("ring_source", {"batches": b, "samples": N, "slots": slots}, {}, "iq")
            buf = ring.output("buffer")
            for s in range(slots):
                buf.ring_select(s).copy_from(np.roll(x, s, axis=0))
            buf.ring_select(0)
            eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
            lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
            rt = js.Runtime([ring] + eng.modules + [lp], graph=True, fuse=True, batch=False)
            rt._keep = [ring] + eng.modules + [lp]
            rt.compute(1 + 2 * slots)
        states.append((lp.state("averagingBuffer").numpy().copy(), lp.state("signalPoints").numpy().copy()))
        rt.destroy()
    assert_bit_equal(states[1][0], states[0][0], "lineplot average")
    assert_bit_equal(states[1][1], states[0][1], "lineplot points")


@pytest.mark.parametrize("b", [1, 15, 16, 17, 63, 64, 65, 130])
def test_lineplot_row_depths_against_the_oracle(js, oracle, b):
    """The Lineplot's ordered batch sum with 16 or 64 rows in flight per thread (waterfall.hip: lineplot_kernel<16> / <64>): every
    batch count around the two depths, with decimation, against the oracle (lineplot/module_impl_native_cpu.cc:80-118)."""
    rng = np.random.default_rng(b)
    x = rng.uniform(0, 1, (b, 1024)).astype(np.float32)
    x[:, ::7] *= np.float32(1e-3)  # sums whose rounding depends on the order
    for averaging, decimation in ((1, 1), (8, 4)):
        t = js.Tensor.from_numpy(x, sample=1, batch=0)
        m = js.Module("lineplot", {"averaging": averaging, "decimation": decimation}, {"signal": t})
        rt = js.Runtime([m], graph=True)
        avg = np.zeros(1024 // decimation, np.float32)
        for cycle in range(3):
            rt.compute()
            oracle.lineplot(avg, x, averaging, decimation)
            assert_bit_equal(m.state("averagingBuffer").numpy(), avg, f"{b} rows, cycle {cycle}")
        rt.destroy()


@pytest.mark.parametrize("b", [5, 40, 70])
def test_lineplot_span_forms_equal_the_per_cycle_kernel(js, oracle, b):
    """lineplot_span_kernel<16, 4> (few rows: four cycles' rows in flight), <16, 1> and <64, 1> behind a cycle-batched spectrum unit:
    spans of 1..7 cycles that wrap a ring of 3 slots, against the per-cycle runtime, bit for bit, after every call."""
    n, slots = 12000, 3  # a tiled length: the LDS-tiled spectrum unit has the span form the Lineplot rides behind
    rng = np.random.default_rng(100 + b)
    xs = [((rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))) * (0.1 + 0.2 * s)).astype(np.complex64) for s in range(slots)]
    traces = []
    for batch in (False, True):
        ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "ring")
        buf = ring.output("buffer")
        for s, x in enumerate(xs):
            buf.ring_select(s).copy_from(x)
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
        lp = js.Module("lineplot", {"averaging": 4}, {"signal": eng.buffer}, "psd")
        rt = js.Runtime([ring] + eng.modules + [lp], fuse=True, graph=True, batch=batch)
        assert rt.batched == batch, rt.units
        trace = []
        for call in (1, 3, 2, 5, 7, 4):
            rt.compute(call)
            trace.append((lp.state("averagingBuffer").numpy().copy(), lp.state("signalPoints").numpy().copy()))
        traces.append(trace)
        rt.destroy()
    for i, (per_cycle, batched) in enumerate(zip(*traces)):
        assert_bit_equal(batched[0], per_cycle[0], f"lineplot average after call {i}")
        assert_bit_equal(batched[1], per_cycle[1], f"lineplot points after call {i}")


@pytest.mark.parametrize("n,b", [(32768, 16), (32768, 256), (131072, 8), (131072, 64), (262144, 4), (262144, 32)])
@pytest.mark.parametrize("provider", ["generic", "fast"])
def test_power_of_two_lengths_on_constant_plans(js, oracle, n, b, provider):
    """Constant plans 10-12 (fft_tiled.hip): 32768 / 131072 / 262144 points -- the smallest transform count that takes them
    (one workgroup per tile) and one large enough for the persistent columns / blocks kernels.  Bit-exact against the oracle
    (pocketfft.hh:1476-1497 order) for provider generic, within 4e-7 for provider fast; rows from both ends of the batch."""
    rng = np.random.default_rng(n + b)
    t = np.arange(n)
    x = (np.exp(2j * np.pi * 777.25 * t / n)[None, :] * (0.5 + rng.uniform(0, 1, (b, 1))) +
         1e-2 * (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n)))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
    rt = js.Runtime(eng.modules, graph=True, fuse=True)
    rt.compute(2)
    got = eng.buffer.numpy()
    rows = np.r_[0:2, b // 2, b - 2:b]
    ref = oracle.spectrum_chain(x[rows], -100.0, 0.0)["range"]
    if provider == "generic":
        assert_bit_equal(got[rows], ref, f"{b} x {n}")
    else:
        err = float(np.max(np.abs(got[rows] - ref)))
        assert err <= 4e-7, err
    rt.destroy()
