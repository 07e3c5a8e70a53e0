"""tests/golden/reference_vectors.npz (frozen outputs of the REFERENCE ITSELF, tools/make_reference_vectors.py) as
runnable cases: load(), the reference tests' own expectations transcribed per case (file:line), and one runner per
side -- run_oracle() for the CPU restatement, run_hip() for the product through the C ABI (ctypes -> libjetstream_hip)."""
import json
import math
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz")
_CACHE = None


def load():
    """{name: {"kind", "params", "source", "ins": [...], "outs": [...]}}"""
    global _CACHE
    if _CACHE is None:
        z = np.load(_PATH)
        manifest = json.loads(bytes(z["manifest"]).decode())
        _CACHE = {}
        for name, c in manifest.items():
            c = dict(c)
            c["ins"] = [z[f"{name}/in{i}"] for i in range(c["cycles"])]
            c["outs"] = [z[f"{name}/out{i}"] for i in range(c["cycles"])]
            _CACHE[name] = c
    return _CACHE


def names(kind=None):
    return sorted(n for n, c in load().items() if kind is None or c["kind"] == kind)


# ------------------------------------------------------------------------------- the reference tests' own assertions
def _tone(signal, channel, freq, sr, first):
    """ToneAmplitude of fm/module_tests.cc:51-65 (F64 correlation against sin / cos)."""
    n = np.arange(first, signal.shape[0], dtype=np.float64)
    ph = 2.0 * math.pi * float(np.float32(freq)) * n / np.float32(sr)
    v = signal[first:, channel].astype(np.float64)
    return float(np.float32(2.0 * math.hypot(float(np.sum(v * np.sin(ph))), float(np.sum(v * np.cos(ph)))) / len(n)))


def expectations(name, outs):
    """Asserts what the reference's own test asserts about these outputs (same tolerances)."""
    near = lambda a, b, tol: abs(complex(a) - complex(b)) <= tol + 1e-12
    if name.startswith("filter_engine_center_"):      # filter_engine/block_tests.cc:584-646
        sign = {"pos": -1.0, "neg": 1.0, "wrapped": -1.0}[name.rsplit("_", 1)[1]]
        o0, o1 = outs
        assert o0.shape == (2,)
        assert abs(o0[0].real - 1.0) <= 1e-5 and abs(o0[1].real - 0.25) <= 1e-5 and abs(o0[1].imag - sign * 0.4330127) <= 1e-5
        assert abs(o1[0].real - 0.25) <= 1e-5 and abs(o1[0].imag + sign * 0.4330127) <= 1e-5
        assert abs(o1[1].real - 1.0) <= 1e-5 and abs(o1[1].imag) <= 1e-5
    elif name == "filter_engine_head_centers":        # :647-724
        o0, o1 = outs
        assert o0.shape == (2, 2)
        for head in range(2):
            sgn = -1.0 if head == 0 else 1.0
            assert near(o0[head, 0], 1.0, 1e-5) and near(o0[head, 1], complex(0.25, sgn * 0.4330127), 1.5e-5)
            assert near(o1[head, 0], complex(0.25, -sgn * 0.4330127), 1.5e-5) and near(o1[head, 1], 1.0, 1e-5)
    elif name == "filter_block_head_centers":         # filter/block_tests.cc:343-400
        o0, o1 = outs
        assert o0.shape == (2, 2)
        for head in range(2):
            sgn = -1.0 if head == 0 else 1.0
            assert abs(o0[head, 0].real) <= 1e-5 and abs(o0[head, 1].real - 0.125) <= 1e-5
            assert abs(o0[head, 1].imag - sgn * 0.21650635) <= 1e-5
            assert abs(o1[head, 0].real + 0.25) <= 1e-5 and abs(o1[head, 0].imag - sgn * 0.4330127) <= 1e-5
            assert abs(o1[head, 1].real + 0.25) <= 1e-5 and abs(o1[head, 1].imag) <= 1e-5
    elif name == "fm_narrow_deemphasis":              # fm/module_tests.cc:204-247
        f = np.float32
        sr = f(240e3)
        inc = f(f(2.0) * f(math.pi) * f(10e3) / sr)
        ref = f(1.0) / (f(2.0) * f(math.pi) * (f(100e3) / sr))
        raw = inc * ref
        alpha = f(1.0) - f(math.exp(float(f(-1.0) / (sr * f(50e-6)))))
        o = outs[0]
        assert abs(o[1] - alpha * raw) <= 1e-5
        assert abs(o[2] - (f(1.0) - (f(1.0) - alpha) * (f(1.0) - alpha)) * raw) <= 1e-5
    elif name == "fm_wide_stereo_multiplex":          # :249-313
        o = outs[0]
        assert o.shape == (8192, 2)
        la, ra = np.float32(0), np.float32(0)
        for v in o[-2048:]:                            # F32 running sums like the test's loop
            la, ra = np.float32(la + v[0]), np.float32(ra + v[1])
        assert abs(la / np.float32(2048) - 0.9 * 0.4) <= 0.01 and abs(ra / np.float32(2048) - 0.9 * -0.2) <= 0.01
    elif name == "fm_wide_tone_separation":           # :315-378
        o, sr, first = outs[0], 240e3, 12000
        assert _tone(o, 0, 15e3, sr, first) > 0.5 and _tone(o, 1, 1e3, sr, first) > 0.75
        assert _tone(o, 0, 1e3, sr, first) < 0.05 and _tone(o, 1, 15e3, sr, first) < 0.05
        assert _tone(o, 0, 19e3, sr, first) < 0.01 and _tone(o, 1, 19e3, sr, first) < 0.01
    elif name.startswith("fm_nonfinite_"):            # :379-418
        o = outs[0]
        assert not np.isfinite(o[2]) and not np.isfinite(o[3]) and np.isfinite(o[4]) and np.isfinite(o[5])
    elif name == "fm_cross_submission":               # :420-483
        f = np.float32
        inc = f(f(2.0) * f(math.pi) * f(10e3) / f(240e3))
        ref = f(1.0) / (f(2.0) * f(math.pi) * (f(100e3) / f(240e3)))
        assert abs(outs[1][0] - inc * ref) <= 0.01


# -------------------------------------------------------------------------------------------------------- CPU oracle
def engine_plan(params, taps_len, heads, signal_size):
    from test_filter_plan import reference_plan
    if "sampleRate" not in params:
        return {"padSize": taps_len - 1, "convolutionSize": signal_size + taps_len - 1, "resample": False,
                "resamplerOffsets": [], "resamplerSize": 0, "resampledSampleRate": 0.0}
    return reference_plan(params["sampleRate"], params["bandwidth"], params["center"], taps_len, heads, signal_size)


def _engine_split(case, x):
    shape = case["params"]["taps_shape"]
    if "taps" in case["params"]:
        return x, np.asarray(case["params"]["taps"], np.complex64).reshape(shape)
    t = int(np.prod(shape))
    return x[:-t], x[-t:].reshape(shape)


def run_oracle(oracle, case):
    from test_filter_plan import reference_plan
    kind, p = case["kind"], case["params"]
    outs, state = [], {}
    if kind == "filter_engine":
        for x in case["ins"]:
            sig, taps = _engine_split(case, x)
            heads = taps.shape[0] if taps.ndim == 2 else 1
            outs.append(oracle.filter_engine_block(sig, taps, engine_plan(p, taps.shape[-1], heads, sig.shape[-1]), state))
    elif kind == "filter":
        for x in case["ins"]:
            xb = np.asarray(x, np.complex64)
            xb = xb.reshape(1, -1) if xb.ndim == 1 else xb
            plan = reference_plan(p["sampleRate"], p["bandwidth"], p["center"], p["taps"], p["heads"], xb.shape[1])
            o = oracle.filter_block(xb, plan, p["sampleRate"], p["bandwidth"], p["center"], p["taps"], state)
            outs.append(o[0] if np.asarray(x).ndim == 1 else o)
    elif kind == "fm":
        lane = oracle.FmLane(p.get("mode", "narrow"), p.get("deemphasis", "none"), p["sampleRate"])
        outs = [lane(x) for x in case["ins"]]
    elif kind == "spectrum_engine":
        outs = [oracle.spectrum_chain(case["ins"][0], p["rangeMin"], p["rangeMax"])["range"]]
    elif kind == "c4":
        lane = oracle.FmLane("wide", "75us", 200e3)
        for x in case["ins"]:
            plan = reference_plan(p["sampleRate"], p["bandwidth"], [0.0], p["taps"], 1, x.shape[1])
            filt = oracle.filter_block(x, plan, p["sampleRate"], p["bandwidth"], [0.0], p["taps"], state)
            audio = lane(filt.reshape(-1))
            n = audio.shape[0] // p["ratio"]
            outs.append(oracle.arithmetic_add(np.ascontiguousarray(audio.reshape(1, n, p["ratio"], 2)), 2).reshape(1, n, 2))
    elif kind == "spectrogram":
        bins = np.zeros(case["ins"][0].shape[1] * p["height"], np.float32)
        for x in case["ins"]:
            oracle.spectrogram(bins, x, p["height"])
            outs.append(bins.reshape(case["outs"][0].shape).copy())
    elif kind == "waterfall":
        bins, st = np.zeros((p["height"], case["ins"][0].shape[1]), np.float32), (0, 0)
        for x in case["ins"]:
            st = oracle.waterfall(bins, st, x, p["height"])
            outs.append(bins.reshape(case["outs"][0].shape).copy())
    elif kind == "lineplot":
        width = case["ins"][0].shape[1] // p["decimation"]
        trace = np.zeros(width, np.float32)
        for c, x in enumerate(case["ins"]):
            oracle.lineplot(trace, x, p["averaging"], p["decimation"])
            pts = case["outs"][c].copy()     # x coordinates are the reference's (i * 2 / (width - 1) - 1): compare y only
            pts.reshape(-1, 2)[:, 1] = trace
            outs.append(pts)
    else:
        raise KeyError(kind)
    return outs


# ---------------------------------------------------------------------------------- the product, through the C ABI
def run_hip(js, case, **rt_flags):
    kind, p = case["kind"], case["params"]
    ins = case["ins"]
    outs = []
    if kind == "filter_engine":
        sig0, taps = _engine_split(case, ins[0])
        src = js.Tensor.from_numpy(np.asarray(sig0, np.complex64), sample=0)
        ft = js.Tensor.from_numpy(np.asarray(taps, np.complex64),
                                  **({"sample": 1, "channel": 0} if taps.ndim == 2 else {"sample": 0}))
        if "sampleRate" in p:
            ft.set_attribute("sampleRate", float(p["sampleRate"]))
            ft.set_attribute("bandwidth", float(p["bandwidth"]))
            c = [float(v) for v in p["center"]]
            ft.set_attribute("center", c if len(c) > 1 else c[0])
        blk = js.FilterEngine(src, ft)
        mods, out_t = blk.modules, blk.buffer
        feed = lambda x: src.copy_from(np.asarray(_engine_split(case, x)[0], np.complex64))
    elif kind == "filter":
        x0 = np.asarray(ins[0])
        axes = {"sample": 0} if x0.ndim == 1 else {"sample": 1, "batch": 0}
        src = js.Tensor.from_numpy(x0, **axes)
        blk = js.Filter(src, p["sampleRate"], p["bandwidth"], p["center"], p["taps"], p["heads"])
        mods, out_t = blk.modules, blk.buffer
        feed = lambda x: src.copy_from(np.asarray(x))
    elif kind == "fm":
        src = js.Tensor.from_numpy(ins[0], sample=0)
        m = js.Module("fm", {"mode": p.get("mode", "narrow"), "deemphasis": p.get("deemphasis", "none"),
                             "sampleRate": p["sampleRate"]}, {"signal": src}, "fm")
        mods, out_t = [m], m.output("signal")
        feed = lambda x: src.copy_from(x)
    elif kind == "spectrum_engine":
        src = js.Tensor.from_numpy(ins[0], sample=1, batch=0)
        eng = js.SpectrumEngine(src, enable_scale=True, range_min=p["rangeMin"], range_max=p["rangeMax"])
        mods, out_t = eng.modules, eng.buffer
        feed = lambda x: src.copy_from(x)
    elif kind == "c4":
        src = js.Tensor.from_numpy(ins[0], sample=1, batch=0)
        flt = js.Filter(src, p["sampleRate"], p["bandwidth"], [0.0], p["taps"], 1)
        sq = js.Module("squeeze_dims", {"axis": 1}, {"buffer": flt.buffer}, "squeeze_head")
        iq = sq.output("buffer").set_axes(batch=0, sample=1)
        fm = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3}, {"signal": iq}, "fm")
        dec = js.Decimator(fm.output("signal"), p["ratio"])
        mods, out_t = flt.modules + [sq, fm] + dec.modules, dec.buffer
        feed = lambda x: src.copy_from(x)
    elif kind in ("spectrogram", "waterfall", "lineplot"):
        src = js.Tensor.from_numpy(ins[0], sample=1, batch=0)
        m = js.Module(kind, dict(p), {"signal": src}, kind)
        rt = js.Runtime([m], **rt_flags)
        for c, x in enumerate(ins):
            if c:
                src.copy_from(x)
            rt.compute(1)
            st = m.state("signalPoints" if kind == "lineplot" else "frequencyBins").numpy()
            outs.append(st.reshape(case["outs"][c].shape))
        rt.destroy()
        return outs
    else:
        raise KeyError(kind)
    rt = js.Runtime(mods, **rt_flags)
    for c, x in enumerate(ins):
        if c:
            feed(x)
        rt.compute(1)
        outs.append(out_t.numpy())
    rt.destroy()
    return outs
