"""The CPU oracle (oracle/jst_oracle.c) against the REFERENCE ITSELF: the reference's own core and native-CPU modules /
blocks compiled in place into oracle/_ref/libref_jetstream.so (oracle/ref_jetstream_build.sh; harness
oracle/ref_jetstream.cc) and driven through Registry::BuildModule / Module::create / Runtime::compute and
Flowgraph::blockCreate / compute.  Every stage of the named path is compared BIT FOR BIT on seeded inputs, and the
numeric vectors the reference's own tests hold are re-run on the reference (so a transcription error would show).

Needs the built library (present in this container and shipped to the GPU box with the snapshot); skipped otherwise --
tests/test_reference_golden.py then still checks the frozen vectors this library produced."""
import math

import numpy as np
import pytest

from oracle import oracle
from oracle import ref_jetstream as rj

pytestmark = pytest.mark.skipif(not rj.available(), reason="oracle/_ref/libref_jetstream.so not built")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype in (np.float64, np.complex128) else np.uint32)


def same(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(bits(a), bits(b)), f"max abs diff {np.nanmax(np.abs(a - b))}"


def cnoise(rng, *shape, scale=1.0):
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * scale).astype(np.complex64)


def run_module(mtype, cfg, inputs, out="buffer", cycles=1):
    """inputs: {port: (array, axes dict, attrs dict)}; returns the output after `cycles` computes (same input)."""
    with rj.RefModule(mtype, cfg) as m:
        for port, spec in inputs.items():
            x, axes = spec[0], (spec[1] if len(spec) > 1 else {})
            m.input(port, x, attrs=(spec[2] if len(spec) > 2 else None), **axes)
        assert m.start() == 0, f"{mtype}: create failed"
        for _ in range(cycles):
            assert m.compute() == 0
        return m.output(out)


# ------------------------------------------------------------------------------------ headline chain, stage by stage
@pytest.mark.parametrize("n", [64, 1000, 4096])
def test_window(n):
    same(run_module("window", {"size": n}, {}, out="window"), oracle.window(n))


@pytest.mark.parametrize("shape,axis", [((4096,), 0), ((3, 64), 1), ((6, 5), 0)])
def test_invert(shape, axis):
    rng = np.random.default_rng(1)
    x = cnoise(rng, *shape)
    got = run_module("invert", {}, {"signal": (x, {"sample": axis})}, out="signal")
    same(got, oracle.invert(x, axis))


def test_multiply_broadcast_and_special_values():
    rng = np.random.default_rng(2)
    a = cnoise(rng, 8, 256)
    b = cnoise(rng, 1, 256)
    a[0, :4] = [complex(np.inf, 1), complex(np.nan, 0), complex(0.0, -0.0), complex(-np.inf, np.inf)]
    b[0, :4] = [complex(0, 0), complex(1, 1), complex(-0.0, 0.0), complex(0, 1)]
    got = run_module("multiply", {}, {"a": (a,), "b": (b,)}, out="product")
    same(got, oracle.multiply(a, b))


@pytest.mark.parametrize("n", [8, 64, 100, 4096, 8050, 65536])
def test_fft_forward_inverse(n):
    rng = np.random.default_rng(n)
    x = cnoise(rng, 3, n)
    for fwd in (True, False):
        got = run_module("fft", {"forward": fwd}, {"signal": (x, {"sample": 1, "batch": 0})}, out="signal")
        same(got, oracle.fft_c2c(x, fwd))


def test_amplitude_range():
    rng = np.random.default_rng(3)
    x = cnoise(rng, 4, 4096, scale=30.0)
    x[0, 0] = 0
    amp = run_module("amplitude", {}, {"signal": (x, {"sample": 1, "batch": 0})}, out="signal")
    same(amp, oracle.amplitude(x, 4096))
    rg = run_module("range", {"min": -100.0, "max": 0.0}, {"signal": (amp,)}, out="signal")
    same(rg, oracle.range_(amp, -100.0, 0.0))


def test_spectrum_engine_block_is_the_composed_chain():
    """spectrum_engine/block_impl.cc wiring on the reference == oracle.spectrum_chain (configs[0] / [1] input form)."""
    rng = np.random.default_rng(1234)
    n, b = 4096, 8
    t = np.arange(n)
    x = np.stack([np.exp(2j * np.pi * (100.25 + r) * t / n) for r in range(b)]).astype(np.complex64)
    x += cnoise(rng, b, n, scale=1e-3)
    with rj.RefFlowgraph() as fg:
        fg.source("src", x, sample=1, batch=0)
        assert fg.block("eng", "spectrum_engine", {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0},
                        {"buffer": "src:signal"}) == 0
        assert fg.state("eng") == 2, "block not created"
        assert fg.compute() == 0
        got = np.array(fg.tensor("eng", "buffer"))
    same(got, oracle.spectrum_chain(x, -100.0, 0.0)["range"])


# ------------------------------------------------------------------------------------------------- Filter-chain stages
def test_pad_unpad():
    rng = np.random.default_rng(4)
    x = cnoise(rng, 3, 2, 50)
    p = run_module("pad", {"size": 7, "axis": 2}, {"unpadded": (x,)}, out="padded")
    same(p, oracle.pad(x, 7, 2))
    with rj.RefModule("unpad", {"size": 7, "axis": 2}) as m:
        m.input("padded", p)
        assert m.run() == 0
        body, tail = oracle.unpad(p, 7, 2)
        same(m.output("unpadded"), body)
        same(m.output("pad"), tail)


@pytest.mark.parametrize("offsets", [None, [0, 3, 37]])
def test_fold(offsets):
    rng = np.random.default_rng(5)
    x = cnoise(rng, 2, 3, 160)
    attrs = {"channelOffsets": (rj.ATTR_VEC_U64, offsets)} if offsets else None
    got = run_module("fold", {"offset": 0 if offsets else 11, "size": 16},
                     {"buffer": (x, {"sample": 2, "channel": 1, "batch": 0}, attrs)})
    if offsets:
        same(got, oracle.fold(x, 2, 16, 0, 1, offsets))
    else:
        same(got, oracle.fold(x, 2, 16, 11))


def test_multiply_constant():
    rng = np.random.default_rng(6)
    x = cnoise(rng, 5, 33)
    c = float(np.float32(1.0) / np.float32(16000))
    got = run_module("multiply_constant", {"constant": c}, {"factor": (x,)}, out="product")
    want = (x.real * np.float32(c) + 1j * (x.imag * np.float32(c))).astype(np.complex64)
    same(got, want)


def test_phase_correction_state_across_computes():
    rng = np.random.default_rng(7)
    x = cnoise(rng, 4, 3, 64)
    inc = [math.remainder(2 * math.pi * o * 159750.0 / 160000.0, 2 * math.pi) for o in (0, 1234, 158000)]
    phases = np.zeros(3, np.float64)
    with rj.RefModule("phase_correction", {"phaseIncrement": 0.0}) as m:
        m.input("signal", x, sample=2, channel=1, batch=0, attrs={"channelPhaseIncrements": (rj.ATTR_VEC_F64, inc)})
        assert m.start() == 0
        for _ in range(3):
            assert m.compute() == 0
            same(m.output("signal"), oracle.phase_correction(x, inc, phases, batch_axis=0, channel_axis=1))


def test_overlap_add_state_across_computes():
    rng = np.random.default_rng(8)
    prev = None
    with rj.RefModule("overlap_add", {}) as m:
        buf = m.input("buffer", cnoise(rng, 4, 2, 40), sample=2, channel=1, batch=0)
        ovl = m.input("overlap", cnoise(rng, 4, 2, 6), sample=2, channel=1, batch=0)
        assert m.start() == 0
        for c in range(3):
            b, o = np.array(buf), np.array(ovl)
            assert m.compute() == 0
            if prev is None:
                prev = np.zeros((1, 2, 6), np.complex64)
            want, prev = oracle.overlap_add(b, o, prev, batch_axis=0)
            same(m.output("buffer"), want)
            buf[...] = cnoise(rng, 4, 2, 40)
            ovl[...] = cnoise(rng, 4, 2, 6)


def test_arithmetic_add():
    rng = np.random.default_rng(9)
    x = cnoise(rng, 3, 25, 4)
    got = run_module("arithmetic", {"operation": "add", "axis": 2}, {"buffer": (x,)})
    same(got, oracle.arithmetic_add(x, 2))


@pytest.mark.parametrize("centers,taps", [([0.0], 101), ([0.3e6, -4e6, 0.0], 251)])
def test_filter_taps(centers, taps):
    got = run_module("filter_taps", {"sampleRate": 20e6, "bandwidth": 2e6, "center": centers, "taps": taps}, {},
                     out="coeffs")
    same(got, oracle.filter_taps(20e6, 2e6, centers, taps))


# ---------------------------------------------------------------------------------------------------------------- FM
def fm_input(rng, n, sr, wide):
    t = np.arange(n) / sr
    audio = 0.5 * np.sin(2 * np.pi * 1e3 * t)
    mpx = 0.9 * (audio + 0.3 * np.sin(2 * np.pi * 3e3 * t) * np.sin(2 * (2 * np.pi * 19e3 * t))) + 0.1 * np.sin(2 * np.pi * 19e3 * t)
    ph = 2 * np.pi * 75e3 * np.cumsum(mpx if wide else audio) / sr
    return (np.exp(1j * ph) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


@pytest.mark.parametrize("mode,deemph", [("narrow", "none"), ("narrow", "50us"), ("narrow", "75us"),
                                         ("wide", "none"), ("wide", "50us"), ("wide", "75us")])
def test_fm_bit_exact_across_submissions_with_nonfinite(mode, deemph):
    rng = np.random.default_rng(1236)
    sr, n = 200e3, 4096
    lane = oracle.FmLane(mode, deemph, sr)
    with rj.RefModule("fm", {"mode": mode, "deemphasis": deemph, "sampleRate": sr}) as m:
        x = fm_input(rng, n, sr, mode == "wide")
        v = m.input("signal", x, sample=0)
        assert m.start() == 0
        for c in range(3):
            if c == 1:
                v[100] = complex(np.nan, 1.0)
                v[2000] = complex(np.inf, 0.0)
            cur = np.array(v)
            assert m.compute() == 0
            same(m.output("signal"), lane(cur))
            v[...] = fm_input(rng, n, sr, mode == "wide")


def test_fm_batched_lanes_layout():
    """[batch, lanes.., sample]: batches continue a lane's stream, other axes are lanes (module_impl_native_cpu.cc:63-90)."""
    rng = np.random.default_rng(11)
    x = cnoise(rng, 3, 2, 512)
    got = run_module("fm", {"mode": "wide", "sampleRate": 200e3}, {"signal": (x, {"sample": 2, "batch": 0})}, out="signal")
    assert got.shape == (3, 2, 512, 2)
    for lane_i in range(2):
        lane = oracle.FmLane("wide", "none", 200e3)
        for b in range(3):
            same(got[b, lane_i], lane(x[b, lane_i]))


# ------------------------------------------------------------------------------------------------------------ blocks
def filter_case(rng, b, s, sr, bw, centers, taps, cycles=2):
    heads = len(centers)
    xs = [cnoise(rng, b, s) for _ in range(cycles)]
    with rj.RefFlowgraph() as fg:
        src = fg.source("src", xs[0], sample=1, batch=0)
        assert fg.block("flt", "filter", {"sampleRate": sr, "bandwidth": bw, "center": centers, "taps": taps,
                                          "heads": heads}, {"signal": "src:signal"}) == 0
        assert fg.state("flt") == 2
        outs = []
        for c in range(cycles):
            src[...] = xs[c]
            assert fg.compute() == 0
            outs.append(np.array(fg.tensor("flt", "buffer")))
    return xs, outs


@pytest.mark.parametrize("b,s,sr,bw,centers,taps", [
    (2, 1500, 20e6, 2e6, [0.0], 101),                    # resample, no translation
    (3, 1950, 20e6, 2e6, [0.3e6, -4.0e6], 51),           # two heads with fold offsets + phase correction
    (2, 1000, 2e6, 1e6, [0.25e6], 21),                   # ratio 2
    (1, 777, 2e6, 0.7e6, [0.0], 33),                     # non-integer ratio: no resampling
])
def test_filter_block_vs_reference(b, s, sr, bw, centers, taps):
    from test_filter_plan import reference_plan
    rng = np.random.default_rng(1235)
    xs, outs = filter_case(rng, b, s, sr, bw, centers, taps)
    plan = reference_plan(sr, bw, centers, taps, len(centers), s)
    state = {}
    for x, want in zip(xs, outs):
        got = oracle.filter_block(x, plan, sr, bw, centers, taps, state)
        same(got, want)


# --------------------------------------------------------------- the reference's own numeric vectors, on the reference
def test_reference_reproduces_its_filter_engine_vectors():
    """filter_engine/block_tests.cc:584-646 -- sanity of the harness: the compiled reference gives its own numbers."""
    for center, sign in ((1.6, -1.0), (-1.6, 1.0), (-7.0, -1.0)):
        with rj.RefFlowgraph() as fg:
            sig = fg.source("sig", np.array([1, -0.5, -0.5, 1], np.complex64), sample=0)
            fg.source("taps", np.array([1, 0, 0], np.complex64), sample=0)
            for k, v in (("sampleRate", 6.0), ("bandwidth", 3.0), ("center", center)):
                fg.set_attr("taps", "signal", k, rj.ATTR_F32, v)
            assert fg.block("eng", "filter_engine", {}, {"signal": "sig:signal", "filter": "taps:signal"}) == 0
            assert fg.compute() == 0
            out = fg.tensor("eng", "buffer")
            assert out.shape == (2,)
            assert abs(out[0] - 1.0) < 1e-5 and abs(out[1] - complex(0.25, sign * 0.4330127)) < 1e-5
            sig[...] = np.array([-0.5, -0.5, 1, -0.5], np.complex64)
            assert fg.compute() == 0
            assert abs(out[0] - complex(0.25, -sign * 0.4330127)) < 1e-5 and abs(out[1] - 1.0) < 1e-5


# ------------------------------------------------------------------------------ the modules either side of the path
@pytest.mark.parametrize("shape,dtype", [("cosine", "CF32"), ("sine", "F32"), ("chirp", "CF32"), ("square", "F32")])
def test_signal_generator_phase_continuity(shape, dtype):
    cfg = {"signalType": shape, "signalDataType": dtype, "sampleRate": 2.0e6, "frequency": 100.25 * 2.0e6 / 4096,
           "amplitude": 1.0, "bufferSize": 4096, "chirpStartFreq": 1e3, "chirpEndFreq": 5e5, "chirpDuration": 0.003}
    state = [0.0, 0.0]
    with rj.RefModule("signal_generator", cfg) as m:
        assert m.start() == 0
        for _ in range(3):
            assert m.compute() == 0
            want, state = oracle.signal(shape, 4096, dtype == "CF32", state, 1.0, cfg["frequency"], 2.0e6, 0.0,
                                        1e3, 5e5, 0.003)
            same(m.output("signal"), want)


def test_agc_tiles():
    rng = np.random.default_rng(12)
    x = cnoise(rng, 2, 5000, scale=3.0)
    x[1, 1024:2048] *= 1e-6
    got = run_module("agc", {"tileSize": 1024}, {"signal": (x, {"sample": 1, "batch": 0})}, out="signal")
    same(got, oracle.agc(x, 1, 1024))


def test_am_two_submissions():
    rng = np.random.default_rng(13)
    lane = oracle.AmLane(0.995)
    with rj.RefModule("am", {"sampleRate": 240e3, "dcAlpha": 0.995}) as m:
        v = m.input("signal", cnoise(rng, 2000), sample=0)
        assert m.start() == 0
        for _ in range(2):
            cur = np.array(v)
            assert m.compute() == 0
            same(m.output("signal"), lane(cur))
            v[...] = cnoise(rng, 2000)


def test_decimator_block():
    rng = np.random.default_rng(14)
    x = cnoise(rng, 3, 400)
    with rj.RefFlowgraph() as fg:
        fg.source("src", x, sample=1, batch=0)
        assert fg.block("dec", "decimator", {"ratio": 4}, {"buffer": "src:signal"}) == 0 and fg.state("dec") == 2
        assert fg.compute() == 0
        got = np.array(fg.tensor("dec", "buffer"))
    same(got, oracle.arithmetic_add(x.reshape(3, 100, 4), 2).reshape(3, 100))


def test_config4_chain_filter_fm_decimator_two_cycles():
    """SURVEY section 8(d) C4 at a CPU-sized length: Filter(20 MS/s -> 200 kS/s, 101 taps) -> FM(wide, 75us) ->
    Decimator(4), two compute cycles (overlap + demodulator state carried), block wiring by the reference."""
    from test_filter_plan import reference_plan
    rng = np.random.default_rng(1236)
    s, sr, bw, taps = 40400, 20e6, 200e3, 101
    t = np.arange(2 * s) / sr
    mpx = 0.9 * 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.1 * np.sin(2 * np.pi * 19e3 * t)
    iq = (np.exp(1j * 2 * np.pi * 75e3 * np.cumsum(mpx) / sr) + cnoise(rng, 2 * s, scale=0.01)).astype(np.complex64)
    xs = [iq[:s].reshape(1, s), iq[s:].reshape(1, s)]
    plan = reference_plan(sr, bw, [0.0], taps, 1, s)
    assert plan["resample"] and plan["resamplerSize"] == 405
    lane, fstate = oracle.FmLane("wide", "75us", 200e3), {}
    with rj.RefFlowgraph() as fg:
        src = fg.source("src", xs[0], sample=1, batch=0)
        assert fg.block("flt", "filter", {"sampleRate": sr, "bandwidth": bw, "center": [0.0], "taps": taps, "heads": 1},
                        {"signal": "src:signal"}) == 0 and fg.state("flt") == 2
        # the wide decoder refuses a channelized input (fm/module_impl.cc:55-58): drop the one-head axis first
        assert fg.block("sq", "squeeze_dims", {"axis": 1}, {"buffer": "flt:buffer"}) == 0 and fg.state("sq") == 2
        fg.set_attr("sq", "buffer", "sampleAxis", rj.ATTR_INDEX, 1)
        assert fg.block("fm", "fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3},
                        {"signal": "sq:buffer"}) == 0 and fg.state("fm") == 2, fg.axes("sq", "buffer")
        assert fg.block("dec", "decimator", {"ratio": 4}, {"buffer": "fm:signal"}) == 0 and fg.state("dec") == 2, \
            (fg.state("dec"), fg.tensor("fm", "signal").shape)
        for x in xs:
            src[...] = x
            assert fg.compute() == 0
            filt = oracle.filter_block(x, plan, sr, bw, [0.0], taps, fstate)        # [1, 1, 404]
            same(np.array(fg.tensor("flt", "buffer")), filt)
            audio = lane(filt.reshape(-1)[: filt.size])                             # [S, 2]
            same(np.array(fg.tensor("fm", "signal")).reshape(-1, 2), audio)
            dec = oracle.arithmetic_add(np.ascontiguousarray(audio.reshape(1, 101, 4, 2)), 2)
            same(np.array(fg.tensor("dec", "buffer")), dec.reshape(1, 101, 2))


# ------------------------------------------------------------- visualization modules' COMPUTE halves (a10, a11, f2)
# Round 5: spectrogram / waterfall / lineplot are compiled in place too (oracle/ref_jetstream_build.sh; their present halves
# are never created: no render window).  State tensors are read as the reference's own tests read them
# (spectrogram/module_tests.cc:22-41, waterfall/module_tests.cc:24-47, lineplot/module_tests.cc:25-44).
def _range_like(rng, b, n, lo=-0.1, hi=1.1):
    """Values around [0, 1] incl. out-of-range, exact bin edges, +-0, and a NaN: what a Range output can hold."""
    x = rng.uniform(lo, hi, (b, n)).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1.0, 0.5, 1.0 / 256, 255.0 / 256, np.nextafter(np.float32(1.0), np.float32(0)), np.nan]
    return x


@pytest.mark.parametrize("b,n,h", [(1024, 4096, 256), (7, 100, 37), (300, 64, 2048)])
def test_spectrogram_state_over_cycles(b, n, h):
    """spectrogram/module_impl_native_cpu.cc:61-87 over three cycles of changing input (decay powf(0.999, B), index
    (U64)(v * H), saturating +0.02): BASELINE config 2's shape first."""
    rng = np.random.default_rng(41)
    xs = [_range_like(rng, b, n) for _ in range(3)]
    bins = np.zeros(n * h, np.float32)
    with rj.RefModule("spectrogram", {"height": h}) as m:
        m.input("signal", xs[0], sample=1, batch=0)
        assert m.start() == 0
        for k, x in enumerate(xs):
            m.write("signal", x)
            assert m.compute() == 0
            oracle.spectrogram(bins, x, h)
            same(m.state("frequencyBins").reshape(-1), bins)
    assert bins.max() > 0.03  # hits accumulated over cycles


def test_spectrogram_leading_sample_axis_and_rank1():
    """The element axis first ([N, B], spectrogram/module_tests.cc:281-329's layouts) and an input without a batch axis."""
    rng = np.random.default_rng(42)
    x = _range_like(rng, 6, 50).T.copy()          # [N=50, B=6]: sampleAxis 0, batchAxis 1
    h = 64
    bins = np.zeros(50 * h, np.float32)
    with rj.RefModule("spectrogram", {"height": h}) as m:
        m.input("signal", x, sample=0, batch=1)
        assert m.run() == 0
        oracle.spectrogram(bins, x, h, batch_axis=1, elem_axis=0)
        same(m.state("frequencyBins").reshape(-1), bins)
    v = _range_like(rng, 1, 333)[0]
    bins = np.zeros(333 * h, np.float32)
    with rj.RefModule("spectrogram", {"height": h}) as m:
        m.input("signal", v, sample=0)
        assert m.run() == 0 and m.compute() == 0
        oracle.spectrogram(bins, v, h)
        oracle.spectrogram(bins, v, h)
        same(m.state("frequencyBins").reshape(-1), bins)


@pytest.mark.parametrize("b,n,h", [(3, 64, 8), (8, 32, 8), (13, 16, 5), (1024, 4096, 512)])
def test_waterfall_ring_over_cycles(b, n, h):
    """waterfall/module_impl_native_cpu.cc:53-78 + ring_state.hh:16-56: fewer batches than rows, as many, MORE (only the
    newest `height` rows land), and config 2's batch on the default height -- four cycles each, the write index too."""
    rng = np.random.default_rng(43)
    bins = np.zeros((h, n), np.float32)
    state = (0, 0)
    with rj.RefModule("waterfall", {"height": h}) as m:
        x0 = rng.standard_normal((b, n)).astype(np.float32)
        m.input("signal", x0, sample=1, batch=0)
        assert m.start() == 0
        for k in range(4):
            x = rng.standard_normal((b, n)).astype(np.float32)
            m.write("signal", x)
            assert m.compute() == 0
            state = oracle.waterfall(bins, state, x, h)
            same(m.state("frequencyBins").reshape(h, n), bins)
            assert m.state_scalar("writeIndex") == state[0]


@pytest.mark.parametrize("b,n,avg,dec", [(16, 65536, 1, 1), (5, 4096, 4, 1), (3, 1000, 2, 4), (1, 64, 8, 3)])
def test_lineplot_trace_over_cycles(b, n, avg, dec):
    """lineplot/module_impl_native_cpu.cc:80-118: left-to-right batch sum per bin, normalisation, fmin / fmax clamp, the moving
    average's two divisions -- config 5's shape first, then averaging and decimation."""
    rng = np.random.default_rng(44)
    width = n // dec
    trace = np.zeros(width, np.float32)
    with rj.RefModule("lineplot", {"averaging": avg, "decimation": dec}) as m:
        x0 = rng.uniform(-0.2, 1.2, (b, n)).astype(np.float32)
        m.input("signal", x0, sample=1, batch=0)
        assert m.start() == 0
        for k in range(3):
            x = rng.uniform(-0.2, 1.2, (b, n)).astype(np.float32)
            m.write("signal", x)
            assert m.compute() == 0
            oracle.lineplot(trace, x, avg, dec)
            pts = m.state("signalPoints").reshape(-1, 2)
            assert pts.shape[0] == width
            same(np.ascontiguousarray(pts[:, 1]), trace)
