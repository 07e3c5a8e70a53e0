"""Pins the CPU oracle (oracle/jst_oracle.c + oracle/_ref) against the reference's own
known-answer tests (tests/golden/reference_kats.json, transcribed with file:line citations)."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def test_fft_kats(oracle):
    for kat in KATS["fft"]:
        if kat["kind"] == "c2c":
            x = np.ones(kat["n"], dtype=np.complex64)
            y = oracle.fft_c2c(x, kat["forward"])
            assert abs(abs(y[0].real) - kat["expect_value"]) <= kat["abs_tol"]
            assert abs(y[0].imag) <= kat["abs_tol"]
            assert np.all(np.abs(y[1:].real) <= kat["abs_tol"]) and np.all(np.abs(y[1:].imag) <= kat["abs_tol"])
        elif kat["kind"] == "c2c_roundtrip":
            n = kat["n"]
            rng = np.random.default_rng(3)
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            back = oracle.fft_c2c(oracle.fft_c2c(x, True), False)
            assert np.max(np.abs(back - n * x)) <= kat["abs_tol"] * n  # unnormalised both ways
        elif not oracle.have_ref():
            pytest.skip("oracle/_ref not built (no /root/reference)")
        elif kat["kind"] == "r2r":
            y = oracle.ref_fft_r2r(np.array(kat["input"], np.float32), forward=kat["forward"])
            assert np.max(np.abs(y - np.array(kat["expect"], np.float32))) <= kat["abs_tol"]
        elif kat["kind"] == "r2c":
            y = oracle.ref_fft_r2c(np.array(kat["input"], np.float32))
            assert np.max(np.abs(y.real - kat["expect_re"])) <= kat["abs_tol"]
            assert np.max(np.abs(y.imag - kat["expect_im"])) <= kat["abs_tol"]


def test_window_kat(oracle):
    for kat in KATS["window"]:
        n = kat["n"]
        w = oracle.window(n)
        i = np.arange(n, dtype=np.float64)
        expect = 0.42 - 0.5 * np.cos(2 * np.pi * i / (n - 1)) + 0.08 * np.cos(4 * np.pi * i / (n - 1))
        assert np.max(np.abs(w.real - expect)) <= kat["abs_tol"]
        assert np.all(w.imag == 0)
    assert oracle.window(1)[0] == 1.0 + 0j


def test_amplitude_kats(oracle):
    for kat in KATS["amplitude"]:
        n = kat["n"]
        if kat["dtype"] == "CF32":
            x = np.full(n, kat["re"] + 1j * kat["im"], dtype=np.complex64)
        else:
            x = np.full(n, kat["re"], dtype=np.float32)
        y = oracle.amplitude(x, n)
        if kat["expect"] == "-inf":
            assert np.all(np.isneginf(y))
        else:
            mag = abs(complex(kat["re"], kat.get("im", 0.0)))
            expect = 20 * math.log10(mag) + 20 * math.log10(1.0 / n)
            assert np.max(np.abs(y - expect)) <= kat["abs_tol"]


def test_approx_log10_is_the_cubic(oracle):
    # helpers.hh:59-74 is a cubic in the frexp mantissa: exact at no point, within 1e-3 of log10
    xs = np.geomspace(1e-30, 1e30, 2001).astype(np.float32)
    err = [abs(oracle.approx_log10(float(x)) - math.log10(float(x))) for x in xs]
    assert max(err) < 2e-3 and max(err) > 1e-5  # genuinely the approximation, not libm


def test_spectrogram_layout_kat(oracle):
    kat = KATS["spectrogram"][0]
    lead = np.array(kat["leading"], np.float32)
    trail = np.array(kat["trailing"], np.float32)
    for height in (4, 256):
        a = np.zeros(3 * height, np.float32)
        b = np.zeros(3 * height, np.float32)
        oracle.spectrogram(a, lead, height, batch_axis=0, elem_axis=1)
        oracle.spectrogram(b, trail, height, batch_axis=1, elem_axis=0)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert a.sum() > 0


def test_spectrogram_index_rule_matches_x86_cast(oracle):
    # The defined-behaviour rule (1 <= f < H) must agree with the reference's literal
    # static_cast<U64>(in * H) followed by 0 < index < H wherever that cast is defined.
    h = 256
    vals = np.array([0.0, 0.0039, 0.00390625, 0.5, 0.999, 0.99999994, 1.0, 1.5, 1e9, 3e18],
                    np.float32)
    for v in vals:
        f = np.float32(v * np.float32(h))
        idx = oracle.lib().jst_oracle_cast_u64(float(f))
        literal = 0 < idx < h
        rule = (f >= 1.0) and (f < np.float32(h))
        assert literal == rule, (v, idx)
    bins = np.zeros(h, np.float32)
    weird = np.array([[-0.5], [np.nan], [np.inf], [-np.inf], [-3.0]], np.float32)
    oracle.spectrogram(bins, weird, h)  # none of these may hit (all fail 0 < index < H on x86-64)
    assert bins.sum() == 0


def test_waterfall_ring_kats(oracle):
    ring_kat, dirty_kat = KATS["waterfall"]
    height = ring_kat["height"]
    width = 3
    ring = np.zeros((height, width), np.float32)
    ref = np.zeros((height, width), np.float32)
    state, ref_write, nxt = (0, 0), 0, 1
    for count in ring_kat["batch_counts"]:
        rows = np.arange(nxt, nxt + count, dtype=np.float32)[:, None] * np.ones((1, width), np.float32)
        nxt += count
        for r in range(count):  # ApplyReferenceWrite, module_tests.cc:161-169
            ref[(ref_write + r) % height] = rows[r]
        ref_write = (ref_write + count % height) % height
        state = oracle.waterfall(ring, state, rows, height)
        assert np.array_equal(ring, ref) and state[0] == ref_write
    chrono = [ring[(state[0] + r) % height, 0] for r in range(height)]
    assert chrono == [nxt - 5, nxt - 4, nxt - 3, nxt - 2, nxt - 1]

    state = (0, 0)
    for step in dirty_kat["steps"]:
        for adv in step["advance"]:
            state = oracle.waterfall_advance(state, adv, dirty_kat["height"])
        assert state == (step["write_index"], step["dirty_rows"])
        if "plan" in step:
            assert list(oracle.waterfall_dirty_plan(state, dirty_kat["height"])) == step["plan"]
        if step.get("clear"):
            state = (state[0], 0)


def test_range_and_invert_and_multiply(oracle):
    s, o = oracle.range_coeffs(-100.0, 0.0)
    assert s == np.float32(0.01) and o == np.float32(1.0)
    assert oracle.range_coeffs(3.0, 3.0) == (0.0, 0.5)
    x = np.array([-100.0, -50.0, 0.0], np.float32)
    y = oracle.range_(x, -100.0, 0.0)
    assert abs(y[1] - 0.5) < 1e-7 and y[0] < 0.02 and y[2] > 0.98
    w = oracle.invert(oracle.window(8))
    assert np.all(np.sign(w.real[1:-1:2]) <= 0) and np.all(w.real[2:-1:2] >= 0)
    a = (np.arange(6, dtype=np.float32).reshape(2, 3) + 1j).astype(np.complex64)
    b = np.array([1 + 1j, 2, 3j], np.complex64).reshape(1, 3)
    assert np.allclose(oracle.multiply(a, b), a * b)


def _agc_case(fn, case):
    """Runs one AGC KAT of tests/golden/reference_kats.json through `fn(x, axis, **config)`."""
    cfgmap = {"tileSize": "tile", "reference": "reference", "epsilon": "epsilon", "minGain": "min_gain",
              "maxGain": "max_gain", "maxGainChange": "max_gain_change"}
    kw = {cfgmap[k]: v for k, v in case["config"].items()}
    axis = case.get("sample_axis", -1)
    if "generate" in case:
        amp = (1 + np.arange(1024) % 16).astype(np.float32)
        x = np.stack([amp, amp, 2 * amp, 2 * amp]).astype(np.float32)
    elif case["dtype"] == "CF32":
        raw = np.array(case["input"], np.float64)
        x = (raw[..., 0].astype(np.float32) + 1j * raw[..., 1].astype(np.float32)).astype(np.complex64)
    else:
        x = np.array(case["input"], np.float32)
    y = fn(x, axis, **kw)
    assert y.shape == x.shape and y.dtype == x.dtype
    if "expect" in case:
        e = np.array(case["expect"], np.float64)
        if case["dtype"] == "CF32":
            e = e[..., 0] + 1j * e[..., 1]
        assert np.max(np.abs(y - e)) <= case["abs_tol"], case["name"]
    if "expect_rows" in case:
        assert np.max(np.abs(y - np.array(case["expect_rows"])[:, None])) <= case["abs_tol"]
    if "expect_abs" in case:
        assert np.all(np.isfinite(y.view(np.float32)))
        assert abs(abs(complex(y[0])) - case["expect_abs"]) <= case["abs_tol"]
    if "expect_signs" in case:
        assert np.all(np.isfinite(y)) and np.array_equal(np.sign(y), np.array(case["expect_signs"], np.float32))
        assert y[0] == 0.0
    if "expect_exact" in case:
        assert np.array_equal(y, np.array(case["expect_exact"], np.float64).astype(np.float32)), case["name"]
    if "expect_ratio" in case:
        s = complex(y[1])
        assert np.isfinite(abs(s)) and s.real > 0 and s.imag < 0
        assert abs(s.imag / s.real - case["expect_ratio"]) <= case["abs_tol"]
    if "expect_rel" in case:
        e = np.array(case["expect_rel"], np.float64)
        assert np.allclose(y.real, e[..., 0], rtol=case["rel_tol"]) and np.allclose(y.imag, e[..., 1], rtol=case["rel_tol"])


def test_agc_kats(oracle):
    for case in KATS["agc"]:
        _agc_case(lambda x, axis, **kw: oracle.agc(x, axis, **kw), case)


def test_cast_scalers(oracle):
    s = KATS["cast"][0]["scalers"]
    for name, npdt in (("I8", np.int8), ("U8", np.uint8), ("I16", np.int16), ("U16", np.uint16),
                       ("I32", np.int32), ("U32", np.uint32)):
        info = np.iinfo(npdt)
        x = np.array([info.min, -1 if info.min < 0 else 1, 0, 1, info.max], npdt)
        y = oracle.cast(x)
        assert y.dtype == np.float32
        assert np.array_equal(y, x.astype(np.float32) / np.float32(s[name]))
        pairs = np.stack([x, x[::-1]], axis=-1)
        z = oracle.cast(pairs, complex_pairs=True)
        assert z.dtype == np.complex64 and z.shape == x.shape
        assert np.array_equal(z.real, y) and np.array_equal(z.imag, y[::-1])
    assert s["CI8"] == 128.0 and s["CI16"] == 32768.0 and s["CU32"] == 2147483648.0


def test_am_kats(oracle):
    """dsp/am/module_tests.cc:52-187, 317-365: the DC blocker's step response decays below 0.1 within 1024
    samples, a modulated carrier yields a varying output, and batches of one lane form ONE sequence."""
    lane = oracle.AmLane(0.995)
    y = lane(np.ones(1024, np.complex64))
    assert y[0] == 1.0 and abs(y[-1]) < 0.1
    assert np.array_equal(y[:4], np.float32(0.995) ** np.arange(4, dtype=np.float32).astype(np.float32)) or \
        np.allclose(y[:4], 0.995 ** np.arange(4), rtol=1e-6)
    t = np.arange(2048) / 240e3
    x = ((1 + 0.5 * np.cos(2 * np.pi * 1e3 * t)) * np.exp(2j * np.pi * 10e3 * t)).astype(np.complex64)
    whole = oracle.AmLane(0.995)(x)
    assert whole.max() - whole.min() > 0.01
    split = oracle.AmLane(0.995)
    assert np.array_equal(np.concatenate([split(x[:700]), split(x[700:])]), whole)
