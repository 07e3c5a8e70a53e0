"""Test cases transcribed from the reference's own module tests (inputs and expectations as written
there, tolerances as written there), run through the C ABI on the device.  One function per
reference TEST_CASE; the docstring gives file and case name."""
import math

import numpy as np
import pytest

from util import run_module

pytestmark = pytest.mark.gpu
FMAX = float(np.finfo(np.float32).max)


def soft_range(value, lo, hi):
    normalized = np.float32((np.float32(value) - np.float32(lo)) / (np.float32(hi) - np.float32(lo)))
    return np.float32(0.5) + np.float32(0.5) * np.float32(math.tanh(np.float32(4.0) * (normalized - np.float32(0.5))))


# ---- core/range/module_tests.cc -----------------------------------------------------------------
def test_range_scales_into_unit_interval(js):
    """Range Module - Scales Into Unit Interval"""
    _, out = run_module(js, "range", {"min": -2.0, "max": 2.0},
                        {"signal": js.Tensor.from_numpy(np.array([-2.0, 0.0, 2.0], np.float32))})
    o = out["signal"]
    assert abs(o[0] - soft_range(-2, -2, 2)) <= 1e-6 and abs(o[1] - 0.5) <= 1e-6 and abs(o[2] - soft_range(2, -2, 2)) <= 1e-6


def test_range_softly_compresses_outliers(js):
    """Range Module - Softly Compresses Outliers"""
    x = np.array([-np.inf, -4.0, 4.0, np.inf], np.float32)
    _, out = run_module(js, "range", {"min": -2.0, "max": 2.0}, {"signal": js.Tensor.from_numpy(x)})
    o = out["signal"]
    assert o[0] == 0.0 and o[3] == 1.0 and 0.0 < o[1] and o[2] < 1.0
    assert abs(o[1] - soft_range(-4, -2, 2)) <= 1e-6 and abs(o[2] - soft_range(4, -2, 2)) <= 1e-6


def test_range_equal_and_reversed_bounds(js):
    """Range Module - Collapses Equal Bounds To Midpoint / Orders Reversed Bounds"""
    x = np.array([-np.inf, -100.0, 100.0, np.inf], np.float32)
    _, out = run_module(js, "range", {"min": 1.0, "max": 1.0}, {"signal": js.Tensor.from_numpy(x)})
    assert np.all(out["signal"] == 0.5)
    _, out = run_module(js, "range", {"min": 1.0, "max": -1.0},
                        {"signal": js.Tensor.from_numpy(np.array([-1.0, 0.0, 1.0], np.float32))})
    o = out["signal"]
    assert abs(o[0] - soft_range(-1, -1, 1)) <= 1e-6 and o[1] == 0.5 and abs(o[2] - soft_range(1, -1, 1)) <= 1e-6


def test_range_rank4_noncontiguous_and_dtype(js):
    """Range Module - Rank 4 Non-Contiguous / Rejects unsupported dtype during validation"""
    storage = (np.arange(2 * 2 * 3 * 2 * 4, dtype=np.float32) + 1).reshape(2, 2, 3, 2, 4)
    t = js.Tensor.from_numpy(storage).slice(0, 1, 2).squeeze_dims(0).permute((1, 0, 3, 2))
    view = storage[1].transpose(1, 0, 3, 2)
    assert tuple(t.shape) == (3, 2, 4, 2) and t.offset != 0
    _, out = run_module(js, "range", {"min": 0.0, "max": 100.0}, {"signal": t})
    ref = np.vectorize(lambda v: soft_range(v, 0.0, 100.0))(view).astype(np.float32)
    assert np.max(np.abs(out["signal"] - ref)) <= 1e-6
    with pytest.raises(js.JetstreamError):
        js.Module("range", {}, {"signal": js.Tensor.from_numpy(np.zeros(16, np.complex64))})


# ---- dsp/invert/module_tests.cc -----------------------------------------------------------------
def test_invert_even_length_alternating_sign(js):
    """Invert Module - Even Length Alternating Sign / Even Length F32 Promotes To CF32"""
    x = np.array([1 + 1j, 2 - 2j, 3 + 3j, 4 - 4j], np.complex64)
    _, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(x)})
    assert np.array_equal(out["signal"], x * np.array([1, -1, 1, -1], np.float32))
    f = np.arange(1, 5, dtype=np.float32)
    _, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(f)})
    assert out["signal"].dtype == np.complex64
    assert np.array_equal(out["signal"], (f * np.array([1, -1, 1, -1], np.float32)).astype(np.complex64))


def _bin_shift(n):
    phase = 2.0 * math.pi * (n // 2) * np.arange(n) / n
    return (np.cos(phase).astype(np.float32) + 1j * np.sin(phase).astype(np.float32)).astype(np.complex64)


def test_invert_odd_length_integer_bin_shift(js):
    """Invert Module - Odd Length Integer Bin Shift / Odd Length F32 Promotes To CF32"""
    v = np.arange(1, 6, dtype=np.float32)
    x = (v - 0.5j * v).astype(np.complex64)
    _, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(x)})
    assert np.max(np.abs(out["signal"] - x * _bin_shift(5))) <= 1e-5
    _, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(v)})
    assert np.max(np.abs(out["signal"] - v.astype(np.complex64) * _bin_shift(5))) <= 1e-5


def test_invert_restarts_per_batch_and_head(js):
    """Invert Module - Leading Batch Restarts For Each Batch / Multi-Head Restarts For Each Head"""
    x = np.array([[(r * 3 + c + 1) + 1j * (c + 1) for c in range(3)] for r in range(2)], np.complex64)
    m, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    assert m.output("signal").axes == {"sample": 1, "batch": 0, "channel": None}
    assert np.max(np.abs(out["signal"] - x * _bin_shift(3)[None, :])) <= 1e-5
    h = np.array([[(hd * 4 + s + 1) * (1 - 1j) for s in range(4)] for hd in range(2)], np.complex64)
    m, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(h, sample=1, channel=0)})
    assert m.output("signal").axes == {"sample": 1, "batch": None, "channel": 0}
    assert np.array_equal(out["signal"], h * np.array([1, -1, 1, -1], np.float32)[None, :])


def test_invert_rank4_opaque_planes_and_strided_view(js):
    """Invert Module - Rank-4 Batched Multi-Head With Opaque Planes / Trailing Batch Strided View"""
    idx = np.arange(2 * 3 * 4 * 2, dtype=np.float32).reshape(2, 3, 4, 2) + 1
    x = (idx + 0.5j * idx).astype(np.complex64)
    m, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(x, sample=2, batch=0, channel=1)})
    sign = np.array([1, -1, 1, -1], np.float32)[None, None, :, None]
    assert np.array_equal(out["signal"], x * sign)
    assert m.output("signal").axes == {"sample": 2, "batch": 0, "channel": 1}
    storage = np.zeros((3, 2, 2), np.complex64)
    for r in range(3):
        for c in range(2):
            storage[r, c, 1] = (r * 2 + c + 1) * (1 - 1j)
    t = js.Tensor.from_numpy(storage).slice(2, 1, 2).squeeze_dims(2).set_axes(sample=0, batch=1)
    view = storage[:, :, 1]
    assert tuple(t.shape) == (3, 2) and t.offset != 0
    m, out = run_module(js, "invert", {}, {"signal": t})
    assert m.output("signal").axes == {"sample": 0, "batch": 1, "channel": None}
    assert np.max(np.abs(out["signal"] - view * _bin_shift(3)[:, None])) <= 1e-5


def test_invert_validation_errors(js):
    """Invert Module - Unsupported DType Error / Missing Or Invalid Signal Metadata Error"""
    with pytest.raises(js.JetstreamError):
        js.Module("invert", {}, {"signal": js.Tensor.from_numpy(np.zeros(3, np.float64), sample=0)})
    z = np.zeros((2, 3), np.complex64)
    with pytest.raises(js.JetstreamError):
        js.Module("invert", {}, {"signal": js.Tensor.from_numpy(z)})                     # no roles
    with pytest.raises(js.JetstreamError):
        js.Module("invert", {}, {"signal": js.Tensor.from_numpy(z, sample=1, batch=2)})  # out of range
    with pytest.raises(js.JetstreamError):
        js.Module("invert", {}, {"signal": js.Tensor.from_numpy(np.zeros(3, np.complex64), sample=0, batch=0)})
    with pytest.raises(js.JetstreamError):
        js.Module("invert", {}, {"signal": js.Tensor.from_numpy(z, sample=1, channel=1)})  # duplicate role


# ---- dsp/amplitude/module_tests.cc --------------------------------------------------------------
def test_amplitude_dc_and_real_signals(js):
    """Amplitude - CF32 DC Signal / F32 Signal (tolerance 0.5 dB as in the reference)"""
    _, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(np.ones(64, np.complex64))})
    assert np.all(np.abs(out["signal"] - 20 * math.log10(1 / 64)) <= 0.5)
    _, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(np.full(128, 2.0, np.float32), sample=0)})
    assert np.all(np.abs(out["signal"] - (20 * math.log10(2.0) + 20 * math.log10(1 / 128))) <= 0.5)


def test_amplitude_metadata_normalization(js):
    """Amplitude - Channel-only Signal / Trailing, Leading Batch Metadata Normalization / Rank 3
    Batched Heads Normalization"""
    five = np.full((5, 2), 5.0, np.float32)
    m, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(five, channel=0, batch=1)})
    assert m.output("signal").axes == {"sample": None, "batch": 1, "channel": 0}
    assert np.all(np.abs(out["signal"] - 20 * math.log10(5.0)) <= 0.1)       # no sample axis: N = 1
    m, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(np.full((5, 3), 5.0, np.float32), sample=0, batch=1)})
    assert np.all(np.abs(out["signal"]) <= 0.1)                              # 20log10(5) + 20log10(1/5)
    m, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(np.full((3, 5), 5.0, np.float32), sample=1, batch=0)})
    assert np.all(np.abs(out["signal"]) <= 0.1)
    x = np.zeros((2, 3, 4), np.float32)
    for b in range(2):
        for h in range(3):
            x[b, h, :] = 4.0 * (b + 1) * (h + 1)
    m, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(x, sample=2, batch=0, channel=1)})
    assert m.output("signal").axes == {"sample": 2, "batch": 0, "channel": 1}
    for b in range(2):
        for h in range(3):
            assert np.all(np.abs(out["signal"][b, h] - 20 * math.log10((b + 1) * (h + 1))) <= 0.1)


def test_amplitude_validation(js):
    """Amplitude - Validation rejects missing or malformed signal metadata / unsupported dtype"""
    z = np.zeros((2, 3), np.float32)
    for axes in ({}, {"sample": 1, "batch": 2}, {"sample": 1, "channel": 1}):
        with pytest.raises(js.JetstreamError):
            js.Module("amplitude", {}, {"signal": js.Tensor.from_numpy(z, **axes)})
    with pytest.raises(js.JetstreamError):
        js.Module("amplitude", {}, {"signal": js.Tensor.from_numpy(np.zeros(4, np.float32), sample=0, batch=0)})
    with pytest.raises(js.JetstreamError):
        js.Module("amplitude", {}, {"signal": js.Tensor.from_numpy(np.zeros(4, np.float64), sample=0)})


# ---- dsp/fold/module_tests.cc -------------------------------------------------------------------
def test_fold_ramp_offset_heads_and_overflow(js):
    """Fold - 1D F32 Ramp / 1D F32 With Offset / 2D F32 Heads Use Channel Offsets / Avoids
    intermediate overflow while averaging"""
    ramp = np.arange(8, dtype=np.float32)
    _, out = run_module(js, "fold", {"offset": 0, "size": 4}, {"buffer": js.Tensor.from_numpy(ramp, sample=0)},
                        outputs=("buffer",))
    assert np.allclose(out["buffer"], [2, 3, 4, 5], atol=1e-5)
    _, out = run_module(js, "fold", {"offset": 2, "size": 4}, {"buffer": js.Tensor.from_numpy(ramp, sample=0)},
                        outputs=("buffer",))
    assert np.allclose(out["buffer"], [4, 5, 2, 3], atol=1e-5)
    heads = np.stack([ramp, 10 + ramp])
    t = js.Tensor.from_numpy(heads, sample=1, channel=0).set_attribute("channelOffsets", [0, 2])
    m, out = run_module(js, "fold", {"offset": 0, "size": 4}, {"buffer": t}, outputs=("buffer",))
    assert np.allclose(out["buffer"], [[2, 3, 4, 5], [14, 15, 12, 13]], atol=1e-5)
    assert m.output("buffer").axes == {"sample": 1, "batch": None, "channel": 0}
    big = np.array([FMAX, FMAX], np.float32)
    _, out = run_module(js, "fold", {"size": 1}, {"buffer": js.Tensor.from_numpy(big, sample=0)}, outputs=("buffer",))
    assert out["buffer"][0] == np.float32(FMAX)            # F64 accumulation: no overflow


def test_fold_rank4_and_trailing_batch(js):
    """Fold - 4D F32 Batched Heads With Opaque Planes / 2D F32 With Trailing Batch"""
    x = np.zeros((2, 2, 8, 2), np.float32)
    for b in range(2):
        for h in range(2):
            for s in range(8):
                for p in range(2):
                    x[b, h, s, p] = 1000 * b + 100 * h + 10 * p + s
    m, out = run_module(js, "fold", {"size": 4}, {"buffer": js.Tensor.from_numpy(x, sample=2, batch=0, channel=1)},
                        outputs=("buffer",))
    assert out["buffer"].shape == (2, 2, 4, 2)
    for b in range(2):
        for h in range(2):
            for s in range(4):
                for p in range(2):
                    assert abs(out["buffer"][b, h, s, p] - (1000 * b + 100 * h + 10 * p + 2 + s)) <= 1e-5
    tb = np.stack([np.arange(8, dtype=np.float32), 10 + np.arange(8, dtype=np.float32)], axis=1)
    m, out = run_module(js, "fold", {"size": 4}, {"buffer": js.Tensor.from_numpy(tb, sample=0, batch=1)},
                        outputs=("buffer",))
    assert out["buffer"].shape == (4, 2) and m.output("buffer").axes == {"sample": 0, "batch": 1, "channel": None}
    assert np.allclose(out["buffer"][:, 0], 2 + np.arange(4), atol=1e-5)
    assert np.allclose(out["buffer"][:, 1], 12 + np.arange(4), atol=1e-5)


# ---- visualization/lineplot/module_tests.cc -----------------------------------------------------
def _lineplot_points(js, array, config, cycles=1, **axes):
    m = js.Module("lineplot", config, {"signal": js.Tensor.from_numpy(array, **axes)})
    rt = js.Runtime([m])
    rt.compute(cycles)
    pts = m.state("signalPoints").numpy().copy()
    rt.destroy()
    return m, pts


def test_lineplot_clamps_amplitudes_before_averaging(js):
    """Lineplot clamps amplitudes before averaging"""
    t = js.Tensor.create("hip", "F32", (2, 4)).set_axes(sample=1, batch=0)
    m = js.Module("lineplot", {"averaging": 2}, {"signal": t})
    rt = js.Runtime([m])
    t.copy_from(np.full((2, 4), -np.inf, np.float32))
    rt.compute(1)
    first = m.state("signalPoints").numpy().copy()
    assert np.all(np.isfinite(first))
    t.copy_from(np.ones((2, 4), np.float32))
    rt.compute(1)
    recovered = m.state("signalPoints").numpy().copy()
    assert np.all(np.isfinite(recovered)) and np.all(recovered[:, 1] > first[:, 1])
    t.copy_from(np.full((2, 4), 2.0, np.float32))
    rt.compute(1)
    assert np.all(m.state("signalPoints").numpy()[:, 1] <= 1.0)
    rt.destroy()


@pytest.mark.parametrize("role", ["sample", "channel"])
def test_lineplot_layouts_are_equivalent(js, role):
    """Lineplot indexes sample and channel batch layouts equivalently / batched decimation uses
    the original row width"""
    leading = np.array([[0.1, 0.2, 0.3, 0.4], [0.2, 0.2, 0.2, 0.2]], np.float32)
    trailing = np.ascontiguousarray(leading.T)
    _, a = _lineplot_points(js, leading, {}, batch=0, **{role: 1})
    _, b = _lineplot_points(js, trailing, {}, batch=1, **{role: 0})
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rows = np.array([[1, 2, 3, 4, 5], [10, 20, 30, 40, 50]], np.float32) / 100.0
    m, pts = _lineplot_points(js, rows, {"decimation": 2}, batch=0, sample=1)
    avg = m.state("averagingBuffer").numpy()
    assert avg.shape == (2,)
    expect = np.clip(np.array([0.11, 0.33], np.float32) * np.float32(1.0 / (0.5 * 2)) - 1.0, -1, 1)
    assert np.allclose(avg, expect, atol=1e-6)             # sums 11 and 33 (x 1/100): columns 0 and 2


# ---- dsp/signal_generator/module_tests.cc -------------------------------------------------------
def test_signal_generator_dc_and_negative_phase(js):
    """Signal Generator - DC F32 / DC CF32 / Negative phase remains normalized"""
    for dtype in ("F32", "CF32"):
        m = js.Module("signal_generator", {"signalType": "dc", "signalDataType": dtype, "amplitude": 0.75,
                                           "dcOffset": 0.25, "bufferSize": 64}, {})
        js.Runtime([m]).compute(1)
        out = m.output("signal").numpy()
        assert np.all(out.real == 1.0) and (dtype == "F32" or np.all(out.imag == 0.0))
    m = js.Module("signal_generator", {"signalType": "cosine", "signalDataType": "CF32", "phase": -math.pi / 2,
                                       "frequency": 0.0, "sampleRate": 1000.0, "bufferSize": 8}, {})
    js.Runtime([m]).compute(2)
    out = m.output("signal").numpy()
    assert np.allclose(out.real, 0.0, atol=1e-6) and np.allclose(out.imag, -1.0, atol=1e-6)
