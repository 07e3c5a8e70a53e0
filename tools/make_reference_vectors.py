#!/usr/bin/env python3
"""Freezes golden vectors PRODUCED BY THE REFERENCE ITSELF (oracle/_ref/libref_jetstream.so: the reference's core and
native-CPU modules / blocks compiled in place, oracle/ref_jetstream_build.sh) into tests/golden/reference_vectors.npz,
so that boxes without /root/reference -- and the -m gpu suite -- can check the oracle and the HIP path against the
reference's own outputs.  Run in the build container:  python tools/make_reference_vectors.py

Every case is (kind, params, inputs per compute cycle, outputs per compute cycle).  The inputs of the cases that
restate one of the reference's own tests are generated the way that test generates them (file:line in `source`);
their expectations are asserted in tests/test_reference_golden.py / tests/test_gpu_reference_golden.py.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_jetstream as rj  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
F32 = np.float32
PI = 3.14159265358979323846  # JST_PI


def cnoise(rng, *shape, scale=1.0):
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * scale).astype(np.complex64)


def polar_f32(phase):
    """std::polar(1.0f, phase) on F32: (cosf, sinf) of the F32 phase."""
    ph = np.asarray(phase, F32)
    return (np.cos(ph.astype(np.float64)).astype(F32) + 1j * np.sin(ph.astype(np.float64)).astype(F32)).astype(np.complex64)


cases = {}
arrays = {}


def record(name, kind, params, ins, outs, source):
    cases[name] = {"kind": kind, "params": params, "cycles": len(ins), "source": source}
    for c, (i, o) in enumerate(zip(ins, outs)):
        arrays[f"{name}/in{c}"] = i
        arrays[f"{name}/out{c}"] = o


# ------------------------------------------------------------------------------------------------- filter_engine KATs
def run_filter_engine(sig_cycles, taps, attrs, taps_axes):
    outs = []
    with rj.RefFlowgraph() as fg:
        sig = fg.source("sig", sig_cycles[0], sample=0)
        fg.source("taps", taps, **taps_axes)
        for k, (kind, v) in attrs.items():
            fg.set_attr("taps", "signal", k, kind, v)
        assert fg.block("eng", "filter_engine", {}, {"signal": "sig:signal", "filter": "taps:signal"}) == 0
        assert fg.state("eng") == 2
        for x in sig_cycles:
            sig[...] = x
            assert fg.compute() == 0
            outs.append(np.array(fg.tensor("eng", "buffer")))
    return outs


sig2 = [np.array([1.0, -0.5, -0.5, 1.0], np.complex64), np.array([-0.5, -0.5, 1.0, -0.5], np.complex64)]
for label, center in (("pos", 1.6), ("neg", -1.6), ("wrapped", -7.0)):
    outs = run_filter_engine(sig2, np.array([1, 0, 0], np.complex64),
                             {"sampleRate": (rj.ATTR_F32, 6.0), "bandwidth": (rj.ATTR_F32, 3.0),
                              "center": (rj.ATTR_F32, center)}, {"sample": 0})
    record(f"filter_engine_center_{label}", "filter_engine",
           {"sampleRate": 6.0, "bandwidth": 3.0, "center": [center], "taps": [1, 0, 0], "taps_shape": [3]},
           sig2, outs, "src/domains/dsp/filter_engine/block_tests.cc:584-646")
outs = run_filter_engine(sig2, np.array([[1, 0, 0], [1, 0, 0]], np.complex64),
                         {"sampleRate": (rj.ATTR_F32, 6.0), "bandwidth": (rj.ATTR_F32, 3.0),
                          "center": (rj.ATTR_VEC_F32, [1.6, -1.6])}, {"sample": 1, "channel": 0})
record("filter_engine_head_centers", "filter_engine",
       {"sampleRate": 6.0, "bandwidth": 3.0, "center": [1.6, -1.6], "taps": [1, 0, 0, 1, 0, 0], "taps_shape": [2, 3]},
       sig2, outs, "src/domains/dsp/filter_engine/block_tests.cc:647-724")
# no metadata: plain convolution, no resampling (CalculateResampleHeuristics bypass, block_impl.cc:49-55)
rng = np.random.default_rng(21)
sigs = [cnoise(rng, 64), cnoise(rng, 64)]
tp = cnoise(rng, 9)
record("filter_engine_no_metadata", "filter_engine", {"taps_shape": [9]}, [np.concatenate([s, tp]) for s in sigs],
       run_filter_engine(sigs, tp, {}, {"sample": 0}), "src/domains/dsp/filter_engine/block_impl.cc:43-56 (bypass)")


# ------------------------------------------------------------------------------------------------------ filter block
def run_filter(cycles, cfg, axes):
    outs = []
    with rj.RefFlowgraph() as fg:
        src = fg.source("src", cycles[0], **axes)
        assert fg.block("flt", "filter", cfg, {"signal": "src:signal"}) == 0 and fg.state("flt") == 2
        for x in cycles:
            src[...] = x
            assert fg.compute() == 0
            outs.append(np.array(fg.tensor("flt", "buffer")))
    return outs


kat = [np.array([1.0, -0.5, -0.5, 1.0], F32), np.array([-0.5, -0.5, 1.0, -0.5], F32)]
cfg = {"sampleRate": 6.0, "bandwidth": 3.0, "taps": 3, "heads": 2, "center": [1.6, -1.6]}
record("filter_block_head_centers", "filter", cfg, kat, run_filter(kat, cfg, {"sample": 0}),
       "src/domains/dsp/filter/block_tests.cc:343-400")
rng = np.random.default_rng(1235)
for label, b, s, cfg in (
        ("three_heads", 3, 1950, {"sampleRate": 20e6, "bandwidth": 2e6, "center": [0.3e6, -4.0e6, 0.0], "taps": 51, "heads": 3}),
        ("c3_shape", 4, 2250, {"sampleRate": 20e6, "bandwidth": 2e6, "center": [0.0], "taps": 251, "heads": 1}),
        ("no_resample", 2, 777, {"sampleRate": 2e6, "bandwidth": 0.7e6, "center": [0.0], "taps": 33, "heads": 1})):
    t = np.arange(s) / cfg["sampleRate"]
    tones = (np.exp(2j * np.pi * 0.3e6 * t) + 0.5 * np.exp(2j * np.pi * 4.0e6 * t)).astype(np.complex64)
    xs = [(tones[None, :] + cnoise(rng, b, s, scale=0.01)).astype(np.complex64) for _ in range(2)]
    record(f"filter_block_{label}", "filter", cfg, xs, run_filter(xs, cfg, {"sample": 1, "batch": 0}),
           "seeded: SURVEY section 8(d) C3 signal form (two tones + AWGN default_rng(1235))")


# ------------------------------------------------------------------------------------------------------------- FM
def run_fm(cycles, cfg, axes=None):
    outs = []
    with rj.RefModule("fm", cfg) as m:
        v = m.input("signal", cycles[0], **(axes or {"sample": 0}))
        assert m.start() == 0
        for x in cycles:
            v[...] = x
            assert m.compute() == 0
            outs.append(m.output("signal"))
    return outs


sr = 240e3
inc = F32(F32(2.0) * F32(PI) * F32(10e3) / F32(sr))
x = polar_f32(np.arange(8, dtype=F32) * inc)
cfg = {"deemphasis": "50us", "sampleRate": sr}
record("fm_narrow_deemphasis", "fm", cfg, [x], run_fm([x], cfg), "src/domains/dsp/fm/module_tests.cc:204-247")

# stereo multiplex, F32 phase accumulation exactly as the test's loop (:262-285)
sr, n = 200e3, 8192
left, right, pilot_off = F32(0.4), F32(-0.2), F32(0.37)
pinc = F32(F32(2.0) * F32(PI) * F32(19e3) / F32(sr))
mscale = F32(F32(2.0) * F32(PI) * F32(75e3) / F32(sr))
phase = F32(0.0)
ph = np.empty(n, F32)
for i in range(n):
    ph[i] = phase
    pp = F32(F32(i) * pinc + pilot_off)
    s_, d_ = F32(F32(0.5) * (left + right)), F32(F32(0.5) * (left - right))
    mpx = F32(F32(0.9) * F32(s_ + F32(d_ * F32(np.sin(np.float64(F32(F32(2.0) * pp)))))) + F32(F32(0.1) * F32(np.sin(np.float64(pp)))))
    phase = F32(phase + F32(mscale * mpx))
x = polar_f32(ph)
cfg = {"mode": "wide", "sampleRate": sr}
record("fm_wide_stereo_multiplex", "fm", cfg, [x], run_fm([x], cfg), "src/domains/dsp/fm/module_tests.cc:249-313")

# tone separation + pilot rejection, F64 phase accumulation as the test's loop (:329-353)
sr, n = 240e3, 24000
tt = np.arange(n, dtype=np.float64) / F32(sr)
l_, r_ = np.sin(2.0 * PI * 15e3 * tt), np.sin(2.0 * PI * 1e3 * tt)
pp = 2.0 * PI * 19e3 * tt + np.float64(F32(0.41))
mpx = 0.9 * (0.5 * (l_ + r_) + 0.5 * (l_ - r_) * np.sin(2.0 * pp)) + 0.1 * np.sin(pp)
car = np.concatenate(([0.0], np.cumsum(2.0 * PI * 75e3 / F32(sr) * mpx)[:-1]))
x = polar_f32(car.astype(F32))
cfg = {"mode": "wide", "sampleRate": sr}
record("fm_wide_tone_separation", "fm", cfg, [x], run_fm([x], cfg), "src/domains/dsp/fm/module_tests.cc:315-378")

for de in ("none", "50us"):
    x = polar_f32(np.arange(6, dtype=F32) * F32(0.2))
    x[2] = complex(np.nan, np.nan)
    cfg = {"deemphasis": de, "sampleRate": 240e3}
    record(f"fm_nonfinite_{de}", "fm", cfg, [x], run_fm([x], cfg), "src/domains/dsp/fm/module_tests.cc:379-418")

inc = F32(F32(2.0) * F32(PI) * F32(10e3) / F32(240e3))
xs = [polar_f32(np.arange(4, dtype=F32) * inc), polar_f32((np.arange(4, dtype=F32) + F32(4)) * inc)]
cfg = {"sampleRate": 240e3}
record("fm_cross_submission", "fm", cfg, xs, run_fm(xs, cfg), "src/domains/dsp/fm/module_tests.cc:420-483")

rng = np.random.default_rng(1236)
for mode, de in (("wide", "75us"), ("narrow", "75us"), ("wide", "none")):
    sr, n = 200e3, 2048
    xs = []
    for c in range(3):
        t = (np.arange(n) + c * n) / sr
        mpx = 0.45 * np.sin(2 * np.pi * 1e3 * t) + 0.1 * np.sin(2 * np.pi * 19e3 * t)
        x = (np.exp(1j * 2 * np.pi * 75e3 * np.cumsum(mpx) / sr) + cnoise(rng, n, scale=0.01)).astype(np.complex64)
        if c == 1:
            x[100] = complex(np.nan, 1.0)
            x[1500] = complex(np.inf, 0.0)
        xs.append(x)
    cfg = {"mode": mode, "deemphasis": de, "sampleRate": sr}
    record(f"fm_seeded_{mode}_{de}", "fm", cfg, xs, run_fm(xs, cfg), "seeded: FM-modulated 1 kHz tone + 19 kHz pilot + AWGN, non-finite samples in cycle 1")

# ------------------------------------------------------------------------------- spectrum chain (configs[0] / [1] form)
n, fs = 4096, 2.0e6
with rj.RefModule("signal_generator", {"signalType": "cosine", "signalDataType": "CF32", "sampleRate": fs,
                                       "frequency": 100.25 * fs / n, "amplitude": 1.0, "bufferSize": n}) as m:
    assert m.run() == 0
    cw = m.output("signal")
rng = np.random.default_rng(1234)
t = np.arange(n)
rows = np.stack([np.exp(2j * np.pi * (100.25 + r) * t / n) for r in range(4)]).astype(np.complex64) + cnoise(rng, 4, n, scale=1e-3)
x = np.concatenate([cw.reshape(1, n), rows]).astype(np.complex64)
with rj.RefFlowgraph() as fg:
    fg.source("src", x, sample=1, batch=0)
    assert fg.block("eng", "spectrum_engine", {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0},
                    {"buffer": "src:signal"}) == 0 and fg.state("eng") == 2
    assert fg.compute() == 0
    record("spectrum_engine_c1_c2_rows", "spectrum_engine", {"rangeMin": -100.0, "rangeMax": 0.0}, [x],
           [np.array(fg.tensor("eng", "buffer"))],
           "SURVEY section 8(d): row 0 = the C1 tone from the reference's signal_generator, rows 1-4 = C2 rows 0-3")

# --------------------------------------------------------------------------------------------- C4 chain, two cycles
s, sr, bw, taps = 20400, 20e6, 200e3, 101
rng = np.random.default_rng(1236)
t = np.arange(2 * s) / sr
mpx = 0.45 * np.sin(2 * np.pi * 1e3 * t) + 0.1 * np.sin(2 * np.pi * 19e3 * t)
iq = (np.exp(1j * 2 * np.pi * 75e3 * np.cumsum(mpx) / sr) + cnoise(rng, 2 * s, scale=0.01)).astype(np.complex64)
xs = [iq[:s].reshape(1, s), iq[s:].reshape(1, s)]
outs = []
with rj.RefFlowgraph() as fg:
    src = fg.source("src", xs[0], sample=1, batch=0)
    assert fg.block("flt", "filter", {"sampleRate": sr, "bandwidth": bw, "center": [0.0], "taps": taps, "heads": 1},
                    {"signal": "src:signal"}) == 0 and fg.state("flt") == 2
    assert fg.block("sq", "squeeze_dims", {"axis": 1}, {"buffer": "flt:buffer"}) == 0 and fg.state("sq") == 2
    fg.set_attr("sq", "buffer", "sampleAxis", rj.ATTR_INDEX, 1)
    assert fg.block("fm", "fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3},
                    {"signal": "sq:buffer"}) == 0 and fg.state("fm") == 2
    assert fg.block("dec", "decimator", {"ratio": 4}, {"buffer": "fm:signal"}) == 0 and fg.state("dec") == 2
    for x in xs:
        src[...] = x
        assert fg.compute() == 0
        outs.append(np.array(fg.tensor("dec", "buffer")))
record("config4_chain", "c4", {"sampleRate": sr, "bandwidth": bw, "taps": taps, "ratio": 4}, xs, outs,
       "SURVEY section 8(d) C4 at 20400 samples per cycle: filter -> squeeze_dims -> fm(wide, 75us) -> decimator(4)")


# ------------------------------------------------------------------------ visualization modules' compute halves (round 5)
def run_visual(mtype, cfg, cycles, state):
    outs = []
    with rj.RefModule(mtype, cfg) as m:
        v = m.input("signal", cycles[0], sample=1, batch=0)
        assert m.start() == 0
        for x in cycles:
            v[...] = x
            assert m.compute() == 0
            outs.append(m.state(state))
    return outs


rng = np.random.default_rng(1237)
b, n, h = 48, 512, 256
xs = []
for c in range(3):
    x = rng.uniform(-0.1, 1.1, (b, n)).astype(F32)
    x[0, :8] = [0.0, -0.0, 1.0, 0.5, 1.0 / 256, 255.0 / 256, np.nextafter(F32(1.0), F32(0)), np.nan]
    xs.append(x)
cfg = {"height": h}
record("spectrogram_three_cycles", "spectrogram", cfg, xs, run_visual("spectrogram", cfg, xs, "frequencyBins"),
       "src/domains/visualization/spectrogram/module_impl_native_cpu.cc:61-87: decay + saturating hits over three cycles, "
       "out-of-range values, exact edges, NaN")
b, n, h = 13, 64, 5   # more batches than rows: only the newest five land
xs = [rng.standard_normal((b, n)).astype(F32) for _ in range(4)]
cfg = {"height": h}
record("waterfall_batches_exceed_height", "waterfall", cfg, xs, run_visual("waterfall", cfg, xs, "frequencyBins"),
       "src/domains/visualization/waterfall/module_impl_native_cpu.cc:53-78 + ring_state.hh:16-56, B > H, four cycles")
b, n = 5, 1000
xs = [rng.uniform(-0.2, 1.2, (b, n)).astype(F32) for _ in range(3)]
cfg = {"averaging": 2, "decimation": 4}
record("lineplot_decimation_averaging", "lineplot", cfg, xs, run_visual("lineplot", cfg, xs, "signalPoints"),
       "src/domains/visualization/lineplot/module_impl_native_cpu.cc:80-118, decimation 4, averaging 2, three cycles")

np.savez_compressed(OUT, manifest=np.frombuffer(json.dumps(cases, sort_keys=True).encode(), np.uint8), **arrays)
print(f"wrote {OUT}: {len(cases)} cases, {os.path.getsize(OUT) / 1024:.0f} KiB")
