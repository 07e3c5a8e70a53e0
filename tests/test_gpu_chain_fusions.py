"""The small-chain fusions of round 5 (cyberether_amd/csrc/modules/chain_fusions.cc) and the one-launch AGC (kernels/agc.hip):
the spectrum_engine block's chain WITH its AGC (spectrum_engine/block_impl.cc:183-197: multiply -> fft -> agc -> amplitude ->
range, one RMS tile per spectrum) on the lengths the reference's multi-fm.yml uses (8000 = 2^6 5^3: the LDS-tiled kernels).
Fused -- `fft_windowed(multiply + fft)` and `agc_amplitude_range(agc + amplitude + range)` (with a Waterfall behind it: its ring
row too), two launches -- the chain must leave
exactly what the module-by-module submission leaves and what the oracle computes (window, pocketfft, F64 AGC, libm-exact
amplitude / range), bit for bit."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def _chain(js, x, fuse, **engine):
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, enable_agc=True, **engine)
    rt = js.Runtime(eng.modules, graph=True, fuse=fuse)
    rt._keep = (src, eng)
    return src, eng, rt


def _oracle(oracle, x):
    n = x.shape[-1]
    st = oracle.spectrum_chain(x)
    level = oracle.agc(st["fft"], axis=-1, tile=n)
    amp = oracle.amplitude(level, n)
    return oracle.range_(amp, -100.0, 0.0), level


@pytest.mark.parametrize("n,b", [(8000, 8), (805, 3), (6000, 1)])
def test_agc_spectrum_chain_fused_equals_unfused_and_the_oracle(js, oracle, n, b):
    rng = np.random.default_rng(n)
    t = np.arange(n)
    x = (np.exp(2j * np.pi * 123.25 * t / n)[None, :] * rng.uniform(0.01, 3.0, (b, 1))
         + 0.05 * (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n)))).astype(np.complex64)
    want, level = _oracle(oracle, x)
    outs = {}
    for fuse in (False, True):
        src, eng, rt = _chain(js, x, fuse)
        units = rt.units
        if fuse:
            assert any(u.startswith("fft_windowed(") for u in units), units
            assert any(u.startswith("agc_amplitude_range(") for u in units), units   # the AGC takes amplitude -> range along
        else:
            assert not any("(" in u for u in units), units
        rt.compute(3)
        outs[fuse] = eng.buffer.numpy()
        assert_bit_equal(eng.agc.output("signal").numpy(), level, f"AGC output (fuse={fuse}): one tile per spectrum")
        x2 = (x * np.float32(0.5)).astype(np.complex64)     # graph replay on new data
        src.copy_from(x2)
        rt.compute(2)
        assert_bit_equal(eng.buffer.numpy(), _oracle(oracle, x2)[0], f"second input (fuse={fuse})")
        rt.destroy()
    assert_bit_equal(outs[True], outs[False], "fused vs module by module")
    assert_bit_equal(outs[True], want, "fused vs the oracle")


def test_amplitude_range_pair_on_its_own(js, oracle):
    """amplitude -> range outside any spectrum chain (an F32 and a CF32 input), generic provider: one unit, same bits."""
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((4, 300)) + 1j * rng.standard_normal((4, 300))).astype(np.complex64)
    z[0, :3] = [0, 1e-30, 1e30]
    for x, axes in ((z, dict(sample=1, batch=0)), (np.abs(z).astype(np.float32), dict(sample=1, batch=0))):
        src = js.Tensor.from_numpy(x, **axes)
        amp = js.Module("amplitude", {}, {"signal": src}, "amp")
        rg = js.Module("range", {"min": -80.0, "max": 10.0}, {"signal": amp.output("signal")}, "rng")
        rt = js.Runtime([amp, rg], graph=True, fuse=True)
        assert rt.units == ["amplitude_range(amp+rng)"], rt.units
        rt.compute(2)
        want = oracle.range_(oracle.amplitude(x, x.shape[-1]), -80.0, 10.0)
        assert_bit_equal(rg.output("signal").numpy(), want, f"amplitude + range on {x.dtype}")
        rt.destroy()


def _filter_input(rng, b, s, sr):
    t = np.arange(b * s).reshape(b, s) / sr
    return (0.6 * np.exp(2j * np.pi * 0.3e6 * t) + 0.3 * np.exp(2j * np.pi * 4.0e6 * t)
            + 0.02 * (rng.standard_normal((b, s)) + 1j * rng.standard_normal((b, s)))).astype(np.complex64)


@pytest.mark.parametrize("case", [
    dict(sr=20e6, bw=2e6, center=[0.0, 3.0e6, -5.0e6], taps=101, s=900, b=2),
    dict(sr=2e6, bw=0.2e6, center=[400e3, -400e3], taps=101, s=7950, b=8),      # multi-fm.yml's Filter: 8050 -> 805 per head
])
def test_filter_tail_with_phase_correction_is_one_unit(js, oracle, case):
    """ifft -> normalize -> phase_correction -> unpad -> overlap_add (filter/block_impl.cc:499-582) as
    `ifft_phase_unpad_overlap`: the correction rides on the inverse transform's last store, the overlap kernel advances the
    phase state and leaves the NEXT cycle's table.  Same output bits and the same F64 phase state as the module-by-module
    submission and the oracle -- also when a fused runtime takes over modules an unfused one has already advanced (the table
    is primed from the state as it stands) and the other way round."""
    sr, bw, center, taps, s, b = (case[k] for k in ("sr", "bw", "center", "taps", "s", "b"))
    rng = np.random.default_rng(99)
    xs = [_filter_input(rng, b, s, sr) for _ in range(7)]
    src = js.Tensor.create("hip", "CF32", (b, s)).set_axes(batch=0, sample=1)
    blk = js.Filter(src, sr, bw, center, taps, len(center))
    assert blk.phase_correction is not None
    state = {}
    done = 0
    for fuse, cycles in ((False, 2), (True, 3), (False, 1), (True, 1)):
        rt = js.Runtime(blk.modules, graph=True, fuse=fuse)
        assert any(u.startswith("ifft_phase_unpad_overlap(") for u in rt.units) == fuse, rt.units
        for _ in range(cycles):
            src.copy_from(xs[done])
            rt.compute()
            ref = oracle.filter_block(xs[done], blk.plan, sr, bw, center, taps, state)
            assert_bit_equal(blk.buffer.numpy(), ref, f"cycle {done} (fuse={fuse})")
            assert_bit_equal(blk.phase_correction.state("phases").numpy(), state["phases"], f"phase state after cycle {done}")
            done += 1
        rt.destroy()


def test_fm_narrow_state_rides_in_the_demodulator_launch(js, oracle):
    """fm (narrow, no de-emphasis): the lane's previous sample is replaced by the thread that read it -- one launch, replayed
    from a hipGraph; the stream over several cycles must equal the oracle's (fm/module_impl_native_cpu.cc:93-139) bit for bit."""
    rng = np.random.default_rng(5)
    b, lanes, s = 8, 2, 805
    src = js.Tensor.create("hip", "CF32", (b, lanes, s)).set_axes(batch=0, channel=1, sample=2)
    fm = js.Module("fm", {"mode": "narrow", "deemphasis": "none", "sampleRate": 200e3}, {"signal": src})
    rt = js.Runtime([fm], graph=True)
    refs = [oracle.FmLane("narrow", "none", 200e3) for _ in range(lanes)]
    for cycle in range(5):
        ph = np.cumsum(rng.uniform(-0.4, 0.4, b * lanes * s)).reshape(b, lanes, s)
        x = (np.exp(1j * ph) * rng.uniform(0.5, 1.5, (b, lanes, s))).astype(np.complex64)
        src.copy_from(x)
        rt.compute()
        got = fm.output("signal").numpy()
        for lane in range(lanes):
            assert_bit_equal(got[:, lane].reshape(-1), np.asarray(refs[lane](x[:, lane, :]), np.float32), f"cycle {cycle} lane {lane}")
    rt.destroy()


@pytest.mark.parametrize("n,b,h", [(805, 8, 512), (6000, 5, 3), (8000, 2, 2)])
def test_agc_chain_takes_the_waterfall_along(js, oracle, n, b, h):
    """agc -> amplitude -> range -> waterfall as ONE launch (ingest_modules.cc TryFuseAgcChain): the block's output, the AGC's
    own output and the Waterfall's ring + cursor after several cycles equal the module-by-module submission and the oracle
    (waterfall/ring_state.hh:16-56) -- also with fewer ring rows than batches (h < b: only the newest rows are kept)."""
    rng = np.random.default_rng(n + h)
    xs = [(rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))).astype(np.complex64) * np.float32(0.1 * (c + 1)) for c in range(4)]
    rings = {}
    for fuse in (False, True):
        src = js.Tensor.from_numpy(xs[0], sample=1, batch=0)
        eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, enable_agc=True)
        wf = js.Module("waterfall", {"height": h}, {"signal": eng.buffer}, "wf")
        rt = js.Runtime(eng.modules + [wf], graph=True, fuse=fuse)
        assert any(u.startswith("agc_amplitude_range_waterfall(") for u in rt.units) == fuse, rt.units
        bins = np.zeros((h, n), np.float32)
        wstate = [0, 0]
        for c, x in enumerate(xs):
            src.copy_from(x)
            rt.compute(1 if c % 2 else 2)
            want, level = _oracle(oracle, x)
            for _ in range(1 if c % 2 else 2):
                wstate = oracle.waterfall(bins, wstate, want, h)
            assert_bit_equal(eng.buffer.numpy(), want, f"block output, input {c} (fuse={fuse})")
            assert_bit_equal(eng.agc.output("signal").numpy(), level, f"AGC output, input {c} (fuse={fuse})")
            assert_bit_equal(wf.state("frequencyBins").numpy().reshape(h, n), bins, f"waterfall ring, input {c} (fuse={fuse})")
        rings[fuse] = (wf.state("frequencyBins").numpy().copy(), wf.state("ringState").numpy().copy())
        rt.destroy()
    assert_bit_equal(rings[True][0], rings[False][0], "ring: fused vs module by module")
    assert np.array_equal(rings[True][1][:2], rings[False][1][:2]), "ring cursor / dirty rows"


def test_duplicate_is_elided_only_when_every_reader_walks_strides(js, oracle):
    """A `slice` block's dense copy (slice -> duplicate) in front of an AGC spectrum engine and an fm (the reference's
    multi-fm.yml): fused, the copy is not made -- `fft_windowed` and the fm read the view -- and the results are those of the
    module-by-module run; with one more reader that wants the dense tensor (here a standalone amplitude) the copy stays and its
    output is valid."""
    rng = np.random.default_rng(77)
    b, heads, n = 4, 2, 805
    ph = np.cumsum(rng.uniform(-0.3, 0.3, b * heads * n)).reshape(b, heads, n)
    x = (np.exp(1j * ph) * rng.uniform(0.2, 1.0, (b, heads, n))).astype(np.complex64)
    results = {}
    for extra_reader in (False, True):
        for fuse in (False, True):
            src = js.Tensor.from_numpy(x, batch=0, channel=1, sample=2)
            sl = js.Module("slice", {"slice": "[:, 1, :]"}, {"buffer": src}, "st.slice")
            dup = js.Module("duplicate", {}, {"buffer": sl.output("buffer")}, "st.duplicate")
            station = dup.output("buffer")
            eng = js.SpectrumEngine(station, enable_scale=True, range_min=-100.0, range_max=0.0, enable_agc=True)
            fm = js.Module("fm", {"mode": "narrow", "deemphasis": "none", "sampleRate": 200e3}, {"signal": station}, "fm")
            mods = [sl, dup] + eng.modules + [fm]
            tap = None
            if extra_reader:
                tap = js.Module("amplitude", {}, {"signal": station}, "tap")
                mods.append(tap)
            rt = js.Runtime(mods, graph=True, fuse=fuse)
            elided = any(u == "st.duplicate(elided)" for u in rt.units)
            assert elided == (fuse and not extra_reader), rt.units
            rt.compute(3)
            results[(extra_reader, fuse)] = (eng.buffer.numpy().copy(), fm.output("signal").numpy().copy())
            if not elided:
                assert_bit_equal(station.numpy(), np.ascontiguousarray(x[:, 1, :]), "the dense copy")
            if tap is not None:
                assert_bit_equal(tap.output("signal").numpy(), oracle.amplitude(np.ascontiguousarray(x[:, 1, :]), n))
            rt.destroy()
    base = results[(False, False)]
    for key, got in results.items():
        assert_bit_equal(got[0], base[0], f"engine output {key}")
        assert_bit_equal(got[1], base[1], f"fm output {key}")
    want, _ = _oracle(oracle, np.ascontiguousarray(x[:, 1, :]))
    assert_bit_equal(base[0], want, "engine output vs the oracle")


def test_an_elided_duplicate_comes_back_with_the_next_plan(js):
    """TryElideDuplicate points the copy's readers at its source; that is a decision of ONE plan.  The same
    modules under a later runtime without fusion must make the copy again -- its output tensor (the slice block's exposed port)
    holds the dense slice, not stale zeros -- and a module that WRITES the source's storage between the duplicate and its
    last reader keeps the copy even when fusing."""
    rng = np.random.default_rng(5)
    b, heads, n = 4, 2, 805
    x = ((rng.standard_normal((b, heads, n)) + 1j * rng.standard_normal((b, heads, n))) * 0.3).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, channel=1, sample=2)
    sl = js.Module("slice", {"slice": "[:, 1, :]"}, {"buffer": src}, "st.slice")
    dup = js.Module("duplicate", {}, {"buffer": sl.output("buffer")}, "st.duplicate")
    fm = js.Module("fm", {"mode": "narrow", "deemphasis": "none", "sampleRate": 200e3}, {"signal": dup.output("buffer")}, "fm")
    mods = [sl, dup, fm]
    rt = js.Runtime(mods, graph=True, fuse=True)
    assert "st.duplicate(elided)" in rt.units, rt.units
    rt.compute(2)
    fused = fm.output("signal").numpy().copy()
    assert not dup.output("buffer").numpy().any()          # the copy was not made: an intermediate nobody inside reads
    rt.destroy()
    rt = js.Runtime(mods, graph=True, fuse=False)          # the SAME modules, planned again without fusion
    assert "st.duplicate" in rt.units and "st.duplicate(elided)" not in rt.units, rt.units
    rt.compute(1)
    assert_bit_equal(dup.output("buffer").numpy(), np.ascontiguousarray(x[:, 1, :]), "the copy is made again")
    rt.destroy()
    # same input twice through fresh demodulator state would differ by the carried phase: compare the per-cycle results instead
    fm2 = js.Module("fm", {"mode": "narrow", "deemphasis": "none", "sampleRate": 200e3}, {"signal": dup.output("buffer")}, "fm2")
    rt = js.Runtime([sl, dup, fm2], graph=True, fuse=False)
    rt.compute(2)
    assert_bit_equal(fm2.output("signal").numpy(), fused, "elided and copied forms demodulate the same samples")
    rt.destroy()
