"""Filter / FM side-chain HIP modules vs the oracle.  Bit-exact for everything computed with
exactly rounded operations (pad, unpad, fold incl. F64 accumulation, overlap_add, arithmetic,
phase_correction, filter_taps); FM uses the device's atan2f/sinf/cosf where the CPU uses libm's,
tolerance 2e-6 (narrow) / 2e-4 (wide) -- the reference's own FM tests use 1e-2
(fm/module_tests.cc:69-203).  KATs follow fold/module_tests.cc:50-391 and
overlap_add/module_tests.cc:46-236."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal, run_module

pytestmark = pytest.mark.gpu


def test_pad_unpad(js, oracle):
    rng = np.random.default_rng(1)
    x = csignal(rng, (3, 2, 50))
    t = js.Tensor.from_numpy(x, sample=2, batch=0, channel=1)
    _, out = run_module(js, "pad", {"size": 14, "axis": 2}, {"unpadded": t}, outputs=("padded",))
    assert_bit_equal(out["padded"], oracle.pad(x, 14, 2))
    _, out = run_module(js, "pad", {"size": 3, "axis": 0}, {"unpadded": t}, outputs=("padded",))
    assert_bit_equal(out["padded"], oracle.pad(x, 3, 0))
    f = rng.standard_normal((4, 9)).astype(np.float32)
    _, out = run_module(js, "unpad", {"size": 4, "axis": -1}, {"padded": js.Tensor.from_numpy(f)},
                        outputs=("unpadded", "pad"))
    body, tail = oracle.unpad(f, 4, -1)
    assert_bit_equal(out["unpadded"], body)
    assert_bit_equal(out["pad"], tail)
    with pytest.raises(js.JetstreamError, match="Size 10 exceeds axis dimension 9"):
        js.Module("unpad", {"size": 10}, {"padded": js.Tensor.from_numpy(f)})
    with pytest.raises(js.JetstreamError, match="out of range"):
        js.Module("pad", {"size": 1, "axis": 5}, {"unpadded": js.Tensor.from_numpy(f)})


def test_fold_kats_and_random(js, oracle):
    # uniform input folds to the same uniform value (fold/module_tests.cc:50-100)
    ones = np.ones(64, np.complex64)
    _, out = run_module(js, "fold", {"size": 16}, {"buffer": js.Tensor.from_numpy(ones)},
                        outputs=("buffer",))
    assert np.all(out["buffer"] == 1)
    # ramp: out[k] = mean_g ramp[k + 16 g]
    ramp = np.arange(64, dtype=np.float32).astype(np.complex64)
    _, out = run_module(js, "fold", {"size": 16}, {"buffer": js.Tensor.from_numpy(ramp)},
                        outputs=("buffer",))
    assert np.allclose(out["buffer"].real, np.arange(16) + 24)
    rng = np.random.default_rng(2)
    x = csignal(rng, (3, 4, 160), scale=1e3)
    for offset in (0, 7, 159, 160):
        t = js.Tensor.from_numpy(x, sample=2, batch=0, channel=1)
        _, out = run_module(js, "fold", {"size": 16, "offset": offset}, {"buffer": t},
                            outputs=("buffer",))
        assert_bit_equal(out["buffer"], oracle.fold(x, 2, 16, offset), f"offset {offset}")
    # per-head offsets through the channelOffsets attribute (filter/block_impl.cc:498)
    t = js.Tensor.from_numpy(x, sample=2, batch=0, channel=1)
    t.set_attribute("channelOffsets", [0, 3, 150, 160])
    _, out = run_module(js, "fold", {"size": 32}, {"buffer": t}, outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.fold(x, 2, 32, 0, 1, [0, 3, 150, 160]))
    with pytest.raises(js.JetstreamError, match="is not a divisor"):
        js.Module("fold", {"size": 7}, {"buffer": js.Tensor.from_numpy(ones)})


def test_overlap_add_state_across_cycles(js, oracle):
    rng = np.random.default_rng(3)
    for batch_axis, shape_b, shape_o, axes in (
        (0, (4, 2, 60), (4, 2, 9), dict(batch=0, channel=1, sample=2)),
        (None, (2, 60), (2, 9), dict(channel=0, sample=1)),
    ):
        tb = js.Tensor.create("hip", "CF32", shape_b).set_axes(**axes)
        to = js.Tensor.create("hip", "CF32", shape_o).set_axes(**axes)
        m = js.Module("overlap_add", {}, {"buffer": tb, "overlap": to})
        rt = js.Runtime([m], graph=True)
        pshape = list(shape_o)
        if batch_axis is not None:
            pshape[batch_axis] = 1
        prev = np.zeros(pshape, np.complex64)
        for cycle in range(4):
            b, o = csignal(rng, shape_b), csignal(rng, shape_o)
            tb.copy_from(b)
            to.copy_from(o)
            rt.compute()
            ref, prev = oracle.overlap_add(b, o, prev, batch_axis)
            assert_bit_equal(m.output("buffer").numpy(), ref, f"cycle {cycle} batch_axis={batch_axis}")
            assert_bit_equal(m.state("previousOverlap").numpy(), prev)


def test_arithmetic_reduction_order(js, oracle):
    rng = np.random.default_rng(4)
    x = csignal(rng, (3, 50, 10), scale=1e4)  # big spread: summation order matters
    t = js.Tensor.from_numpy(x, batch=0, sample=1)
    m, out = run_module(js, "arithmetic", {"operation": "add", "axis": 2}, {"buffer": t},
                        outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.arithmetic_add(x, 2))
    m, out = run_module(js, "arithmetic", {"operation": "add", "axis": 1, "squeeze": True},
                        {"buffer": js.Tensor.from_numpy(x, batch=0, sample=2)}, outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.arithmetic_add(x, 1).reshape(3, 10))
    assert m.output("buffer").axes == {"sample": 1, "batch": 0, "channel": None}
    f = (rng.standard_normal((6, 40)) * 1e3).astype(np.float32)
    # strided view: decimator wiring reshapes [.., S] -> [.., S/r, r] and sums the last axis
    t = js.Tensor.from_numpy(f, batch=0, sample=1)
    t.reshape((6, 4, 10))
    _, out = run_module(js, "arithmetic", {"operation": "add", "axis": -1}, {"buffer": t},
                        outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.arithmetic_add(f.reshape(6, 4, 10), 2))
    with pytest.raises(js.JetstreamError, match="Invalid operation"):
        js.Module("arithmetic", {"operation": "xor"}, {"buffer": t})


@pytest.mark.parametrize("taps,heads", [(1, 1), (51, 1), (251, 3), (1001, 2)])
def test_filter_taps(js, oracle, taps, heads):
    center = [0.0, 0.3e6, -4.0e6][:heads]
    m, out = run_module(js, "filter_taps", {"sampleRate": 20e6, "bandwidth": 2e6, "center": center,
                                            "taps": taps}, {}, outputs=("coeffs",))
    assert_bit_equal(out["coeffs"], oracle.filter_taps(20e6, 2e6, center, taps))
    assert m.output("coeffs").axes == {"sample": 1, "batch": None, "channel": 0}
    with pytest.raises(js.JetstreamError, match="must be odd"):
        js.Module("filter_taps", {"taps": 100}, {})


def test_phase_correction_state(js, oracle):
    rng = np.random.default_rng(5)
    x = csignal(rng, (5, 3, 40))
    t = js.Tensor.create("hip", "CF32", x.shape).set_axes(batch=0, channel=1, sample=2)
    inc = [0.1, -2.5, 7.0]
    t.set_attribute("channelPhaseIncrements", inc)
    m = js.Module("phase_correction", {}, {"signal": t})
    rt = js.Runtime([m], graph=True)
    phases = np.zeros(3, np.float64)
    for cycle in range(3):
        x = csignal(rng, x.shape)
        t.copy_from(x)
        rt.compute()
        ref = oracle.phase_correction(x, inc, phases, batch_axis=0, channel_axis=1)
        assert_bit_equal(m.output("signal").numpy(), ref, f"cycle {cycle}")
        assert np.array_equal(m.state("phases").numpy(), phases)
    # scalar increment, no channel axis
    y = csignal(rng, (4, 32))
    t2 = js.Tensor.from_numpy(y, batch=0, sample=1)
    _, out = run_module(js, "phase_correction", {"phaseIncrement": 0.25}, {"signal": t2})
    assert_bit_equal(out["signal"], oracle.phase_correction(y, 0.25, np.zeros(1), batch_axis=0))


def fm_signal(rng, n, sr, dev, wide):
    t = np.arange(n) / sr
    audio = np.sin(2 * np.pi * 1e3 * t)
    if wide:
        audio = 0.45 * audio + 0.1 * np.sin(2 * np.pi * 19e3 * t) + \
            0.2 * np.sin(2 * np.pi * 400 * t) * np.sin(2 * np.pi * 38e3 * t)
    phase = 2 * np.pi * dev * np.cumsum(audio) / sr
    x = np.exp(1j * phase) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


@pytest.mark.parametrize("mode,deemph", [("narrow", "none"), ("narrow", "75us"), ("wide", "none"), ("wide", "50us")])
def test_fm_against_oracle_across_submissions(js, oracle, mode, deemph):
    """BIT-EXACT (round 2; 2e-6 / 2e-4 absolute before): atan2f / sinf / cosf inside the kernels are the
    restatements of the host libm's routines (kernels/libm_float.hh, swept against libm.so.6 on every float by
    tests/test_libm_float.py), so the discriminator, the pilot tracker and every recurrence behind them see the
    operands the reference's CPU path sees (fm/module_impl_native_cpu.cc:93,123-139)."""
    rng = np.random.default_rng(6)
    sr, lanes, batches, samples = 240e3, 3, 2, 1500
    wide = mode == "wide"
    t = js.Tensor.create("hip", "CF32", (batches, lanes, samples)).set_axes(
        batch=0, sample=2, **({} if wide else {"channel": 1}))
    m = js.Module("fm", {"mode": mode, "deemphasis": deemph, "sampleRate": sr}, {"signal": t})
    rt = js.Runtime([m])
    refs = [oracle.FmLane(mode, deemph, sr) for _ in range(lanes)]
    for cycle in range(3):
        x = np.stack([fm_signal(rng, batches * samples, sr, 75e3 if wide else 5e3, wide)
                      .reshape(batches, samples) for _ in range(lanes)], axis=1)
        if cycle == 1:
            x[0, 1, 10] = complex(np.nan, 0)  # non-finite sample: NaN out, state not poisoned
        t.copy_from(x)
        rt.compute()
        got = m.output("signal").numpy()
        for lane in range(lanes):
            ref = refs[lane](x[:, lane, :])
            g = got[:, lane].reshape(-1, 2) if wide else got[:, lane].reshape(-1)
            assert np.array_equal(np.isnan(g), np.isnan(ref))
            ok = ~np.isnan(ref)
            assert np.array_equal(g[ok].view(np.uint32), np.asarray(ref, np.float32)[ok].view(np.uint32)), \
                (cycle, lane, float(np.max(np.abs(g[ok] - ref[ok]))))
    if cycle == 0:
        assert got.reshape(-1)[0] == 0.0  # first-ever sample demodulates to exactly 0


def test_fm_reference_kats(js):
    # constant phase -> 0; linear phase ramp -> dphi * ref (fm/module_tests.cc:69-203, tol 1e-2)
    sr, n = 240e3, 512
    const = np.ones(n, np.complex64)
    _, out = run_module(js, "fm", {"sampleRate": sr}, {"signal": js.Tensor.from_numpy(const)})
    assert np.max(np.abs(out["signal"])) < 1e-6
    dphi = 0.1
    ramp = np.exp(1j * dphi * np.arange(n)).astype(np.complex64)
    _, out = run_module(js, "fm", {"sampleRate": sr}, {"signal": js.Tensor.from_numpy(ramp)})
    ref = 1.0 / (2 * np.pi * (100e3 / sr))
    assert out["signal"][0] == 0 and np.max(np.abs(out["signal"][1:] - dphi * ref)) < 1e-2
    with pytest.raises(js.JetstreamError, match="at least 200 kHz"):
        js.Module("fm", {"mode": "wide", "sampleRate": 100e3}, {"signal": js.Tensor.from_numpy(ramp)})


def test_signal_generator_cosine_continuity(js, oracle):
    """CW tone generator (the input of BASELINE config 1): bit-exact vs the oracle, phase carried
    across submissions (signal_generator/module_impl_native_cpu.cc:159-163,222-231)."""
    n, fs = 4096, 2.0e6
    f0 = 100.25 * fs / n
    for dtype in ("CF32", "F32"):
        m = js.Module("signal_generator", {"signalType": "cosine", "signalDataType": dtype,
                                           "sampleRate": fs, "frequency": f0, "amplitude": 0.75,
                                           "dcOffset": 0.1, "phase": 7.0, "bufferSize": n}, {})
        rt = js.Runtime([m], graph=True)
        ph = 7.0
        for cycle in range(3):
            rt.compute()
            ref, ph = oracle.signal_cosine(n, 0.75, f0, fs, 0.1, ph)
            got = m.output("signal").numpy()
            if dtype == "F32":
                ref = np.ascontiguousarray(ref.real)
            assert_bit_equal(got, ref, f"{dtype} cycle {cycle}")
    with pytest.raises(js.JetstreamError, match="Invalid signal type 'warble'"):
        js.Module("signal_generator", {"signalType": "warble"}, {})
    with pytest.raises(js.JetstreamError, match="Frequency .* must be within the supported range"):
        js.Module("signal_generator", {"signalType": "square", "frequency": -5.0}, {})


@pytest.mark.parametrize("shape", ["sine", "square", "triangle", "sawtooth", "dc", "chirp"])
@pytest.mark.parametrize("dtype", ["F32", "CF32"])
def test_signal_generator_waveforms(js, oracle, shape, dtype):
    """Every deterministic waveform (module_impl_native_cpu.cc:192-375) bit-exact vs the oracle,
    oscillator phase / chirp time carried across submissions.  The negative-frequency sine covers
    the downward phase walk."""
    n, fs = 3000, 48000.0
    cfg = {"signalType": shape, "signalDataType": dtype, "sampleRate": fs, "amplitude": 0.6, "dcOffset": -0.05,
           "phase": 1.25, "bufferSize": n, "frequency": 997.3,
           "chirpStartFreq": 100.0, "chirpEndFreq": 9000.0, "chirpDuration": 0.11}
    if shape == "sine" and dtype == "CF32":
        cfg["frequency"] = -4321.0
    m = js.Module("signal_generator", cfg, {})
    rt = js.Runtime([m], graph=True)
    state = [1.25, 0.0]
    for cycle in range(3):   # 9000 samples: the chirp wraps its 0.11 s sweep once
        rt.compute()
        ref, state = oracle.signal(shape, n, dtype == "CF32", state, 0.6, cfg["frequency"], fs, -0.05,
                                   100.0, 9000.0, 0.11)
        assert_bit_equal(m.output("signal").numpy(), ref, f"{shape} {dtype} cycle {cycle}")


@pytest.mark.parametrize("dtype", ["F32", "CF32"])
def test_signal_generator_noise_statistics(js, dtype):
    """Noise is seeded from std::random_device in the reference, so only the distribution is
    specified (signal_generator/module_tests.cc checks mean and variance the same way)."""
    n = 1 << 18
    m = js.Module("signal_generator", {"signalType": "noise", "signalDataType": dtype, "amplitude": 2.0,
                                       "noiseVariance": 0.25, "dcOffset": 0.5, "bufferSize": n}, {})
    rt = js.Runtime([m], graph=True)
    rt.compute()
    a = m.output("signal").numpy().copy()
    rt.compute()
    b = m.output("signal").numpy()
    assert not np.array_equal(a, b)                      # the stream advances between submissions
    re = a.real if dtype == "CF32" else a
    assert abs(re.mean() - 0.5) < 0.01 and abs(re.std() - 1.0) < 0.01   # scale = 2 * sqrt(0.25)
    if dtype == "CF32":
        assert abs(a.imag.mean()) < 0.01 and abs(a.imag.std() - 1.0) < 0.01
        assert abs(np.corrcoef(a.real, a.imag)[0, 1]) < 0.01
    z = (re - 0.5)
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.1   # gaussian moments


@pytest.mark.parametrize("mode,deemph", [("wide", "none"), ("wide", "75us"), ("narrow", "50us")])
def test_fm_wide_wavefront_pipeline_equals_serial_walk(js, switch, mode, deemph):
    """The stage-split wide decoder (fm_wide_kernel: lane pipelines over DPP for the one-poles and the
    biquad cascades) against the one-thread-per-lane walk of the same recurrences (JST_FM_SERIAL=1):
    identical bits, over submissions, with non-finite samples travelling through as bubbles."""
    rng = np.random.default_rng(8)
    sr, lanes, batches, samples = 240e3, 2, 3, 2311   # several LDS chunks with a ragged tail
    outs = {}
    for variant in ("serial", "pipeline"):
        switch("JST_FM_SERIAL", "1" if variant == "serial" else None)
        rng = np.random.default_rng(8)
        t = js.Tensor.create("hip", "CF32", (batches, lanes, samples)).set_axes(batch=0, sample=2)
        m = js.Module("fm", {"mode": mode, "deemphasis": deemph, "sampleRate": sr}, {"signal": t})
        rt = js.Runtime([m])
        got = []
        for cycle in range(3):
            x = np.stack([fm_signal(rng, batches * samples, sr, 75e3, mode == "wide").reshape(batches, samples)
                          for _ in range(lanes)], axis=1)
            if cycle >= 1:
                x[0, 1, 10] = complex(np.nan, 0)
                x[1, 0, 0] = complex(np.inf, 1)
                x[2, 1, samples - 1] = complex(0, -np.inf)
                x[1, 1, 100:103] = complex(np.nan, np.nan)
            t.copy_from(x)
            rt.compute()
            got.append(m.output("signal").numpy().copy())
        outs[variant] = got
        rt.destroy()
    for cycle in range(3):
        a, b = outs["serial"][cycle], outs["pipeline"][cycle]
        assert np.array_equal(np.isnan(a), np.isnan(b)), cycle
        assert np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)]), cycle
