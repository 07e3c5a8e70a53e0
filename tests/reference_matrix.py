"""A differential matrix over the reference's module tests: the (module, config, input layout) tuples the reference's own
`module_tests.cc` files exercise -- dense, batched (leading / trailing batch axis), multi-head (channelAxis), rank 3, strided and
offset views made the way those tests make them (a storage tensor, `slice` with an index or a range, `permute`,
`broadcastTo`), every supported sample type, and the malformed inputs their validation sections feed -- written as DATA, so
that the same tuple can be driven
  * through the REFERENCE (oracle/_ref/libref_jetstream.so: Registry::BuildModule -> Module::create -> Runtime::compute,
    the calls TestContext::run makes, src/testing.cc:123-128): `run_reference`, which tools/make_reference_matrix.py uses to
    freeze (Result code, output, output axes) per case into tests/golden/reference_matrix.npz, and
  * through the product (ctypes -> C ABI -> HIP): `run_hip`, compared on the GPU case by case -- same accept / reject
    decision, same output bits -- by tests/test_gpu_reference_matrix.py.
Every case names the reference test section(s) whose scenario it replays (`cite`).  Test infrastructure."""
from __future__ import annotations

import json
import os
import zlib

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_matrix.npz")
D = "src/domains/"

CASES = []


def case(name, module, config, inputs, out, cite, cycles=1):
    """inputs: {port: dict(shape=storage shape, dtype=numpy dtype name or "CI8" ..., views=[(op, ...)], axes={sample, batch,
    channel}, fill="rand" | "ramp" | "zeros" | "tone")}"""
    assert name not in {c["name"] for c in CASES}, name
    CASES.append({"name": name, "module": module, "config": config, "inputs": inputs, "out": out, "cite": cite, "cycles": cycles})


def tensor(shape, dtype="complex64", views=(), fill="rand", **axes):
    return {"shape": list(shape), "dtype": dtype, "views": [list(v) for v in views], "axes": axes, "fill": fill}


# ---- layouts the reference's tests build (the view ops run on both sides) ------------------------------------------------
def layouts(n, b=3, c=2):
    """(tag, tensor kwargs) for a transform / element axis of n samples."""
    return [
        ("dense", dict(shape=(n,), sample=0)),
        ("batch_leading", dict(shape=(b, n), sample=1, batch=0)),
        ("batch_trailing", dict(shape=(n, b), sample=0, batch=1)),
        ("heads", dict(shape=(c, n), sample=1, channel=0)),
        ("rank3_batch_heads", dict(shape=(b, c, n), sample=2, batch=0, channel=1)),
        # fft/module_tests.cc:596-645: storage [3, n, c], index 1 of axis 0, transposed -> [c, n], offset != 0, not contiguous
        ("heads_strided_offset", dict(shape=(3, n, c), views=[("select", 0, 1), ("permute", 1, 0)], sample=1, channel=0)),
        # fft/module_tests.cc:647-700: storage [n, b, 2], index 1 of the last axis -> [n, b] with stride 2, trailing batch
        ("batch_trailing_strided", dict(shape=(n, b, 2), views=[("select", 2, 1)], sample=0, batch=1)),
        # a range with a step along the batch axis (amplitude/module_tests.cc:475-580 style non-contiguous rows)
        ("batch_stepped", dict(shape=(2 * b, n), views=[("range", 0, 1, 2 * b, 2)], sample=1, batch=0)),
    ]


# ---- FFT (dsp/fft/module_tests.cc) ------------------------------------------------------------------------------------------
for n in (8, 60):
    for fwd in (True, False):
        for tag, kw in layouts(n):
            case(f"fft_c2c_{'fwd' if fwd else 'inv'}_{n}_{tag}", "fft", {"forward": fwd}, {"signal": tensor(**kw)}, "signal",
                 D + "dsp/fft/module_tests.cc:52-147,445-536,596-700,802-852 (c2c, layouts)")
for n in (4, 5, 16):
    for fwd in (True, False):
        for cplx in (False, True):
            for tag, kw in layouts(n)[:3] + layouts(n)[6:7]:
                case(f"fft_real_{'fwd' if fwd else 'inv'}_{'r2c' if cplx else 'fftpack'}_{n}_{tag}", "fft",
                     {"forward": fwd, "complexOutput": cplx}, {"signal": tensor(dtype="float32", **kw)}, "signal",
                     D + "dsp/fft/module_tests.cc:149-443,854-897 (FFTPACK r2r, r2c, inverse, edge lengths, batched strided)")
case("fft_invalid_empty_f32", "fft", {}, {"signal": tensor(shape=(0,), dtype="float32", sample=0)}, "signal", D + "dsp/fft/module_tests.cc:702-752")
case("fft_invalid_rank2_without_axes", "fft", {}, {"signal": tensor(shape=(2, 4))}, "signal", D + "dsp/fft/module_tests.cc:702-752")
case("fft_invalid_duplicate_roles", "fft", {}, {"signal": tensor(shape=(4,), sample=0, batch=0)}, "signal", D + "dsp/fft/module_tests.cc:702-752")
case("fft_invalid_batch_out_of_range", "fft", {}, {"signal": tensor(shape=(2, 3), sample=1, batch=2)}, "signal", D + "dsp/fft/module_tests.cc:702-752")
case("fft_invalid_sample_and_channel_same_axis", "fft", {}, {"signal": tensor(shape=(2, 3), sample=1, channel=1)}, "signal", D + "dsp/fft/module_tests.cc:702-752")
case("fft_invalid_f64", "fft", {}, {"signal": tensor(shape=(4,), dtype="float64", sample=0)}, "signal", D + "dsp/fft/module_tests.cc:702-752")

# ---- Cast (core/cast/module_tests.cc) -----------------------------------------------------------------------------------------
for src in ("CI8", "CI16", "CU8", "CU16", "int8", "int16", "uint8", "float32", "complex64"):
    for dst in ("CF32", "F32", "CI8", "CI16", "I16"):
        case(f"cast_{src}_to_{dst}", "cast", {"outputType": dst}, {"buffer": tensor(shape=(3, 16), dtype=src, sample=1, batch=0)}, "buffer",
             D + "core/cast/module_tests.cc (type matrix: every input type x requested output type, accept / reject and values)")
for tag, kw in layouts(16)[4:8]:
    case(f"cast_CI16_to_CF32_{tag}", "cast", {"outputType": "CF32"}, {"buffer": tensor(dtype="CI16", **kw)}, "buffer",
         D + "core/cast/module_tests.cc (non-contiguous and offset inputs)")

# ---- Signal generator (dsp/signal_generator/module_tests.cc) -------------------------------------------------------------------
for wave in ("sine", "cosine", "square", "triangle", "sawtooth", "dc", "chirp"):
    for dt in ("F32", "CF32"):
        for k, extra in enumerate(({}, {"frequency": 12345.678, "amplitude": 0.37, "phase": 1.1, "dcOffset": -0.2})):
            cfg = {"signalType": wave, "signalDataType": dt, "sampleRate": 1.0e6, "bufferSize": 257, **extra}
            if wave == "chirp":
                cfg.update({"chirpStartFreq": 1000.0, "chirpEndFreq": 90000.0, "chirpDuration": 0.0004})
            case(f"siggen_{wave}_{dt}_{k}", "signal_generator", cfg, {}, "signal",
                 D + "dsp/signal_generator/module_tests.cc (waveform x data type, parameters, phase continuity over three submissions)", cycles=3)
case("siggen_invalid_type", "signal_generator", {"signalType": "nope", "bufferSize": 8}, {}, "signal", D + "dsp/signal_generator/module_tests.cc (validation)")
case("siggen_invalid_zero_buffer", "signal_generator", {"bufferSize": 0}, {}, "signal", D + "dsp/signal_generator/module_tests.cc (validation)")
case("siggen_invalid_dtype", "signal_generator", {"signalDataType": "CI8", "bufferSize": 8}, {}, "signal", D + "dsp/signal_generator/module_tests.cc (validation)")

# ---- Amplitude (dsp/amplitude/module_tests.cc) -----------------------------------------------------------------------------------
for dt in ("complex64", "float32"):
    for tag, kw in layouts(32):
        case(f"amplitude_{dt}_{tag}", "amplitude", {}, {"signal": tensor(dtype=dt, **kw)}, "signal",
             D + "dsp/amplitude/module_tests.cc:50-128,475-580 (constant and noise inputs, rank-4 / non-contiguous layouts)")
case("amplitude_zeros_give_minus_inf", "amplitude", {}, {"signal": tensor(shape=(2, 16), fill="zeros", sample=1, batch=0)}, "signal",
     D + "dsp/amplitude/module_tests.cc:419-473")

# ---- Multiply (core/multiply/module_tests.cc:84-420) -----------------------------------------------------------------------------
M = D + "core/multiply/module_tests.cc:84-420 (broadcast forms, types, mismatches)"
for tag, a, b in (("same", (3, 8), (3, 8)), ("row_broadcast", (3, 8), (1, 8)), ("col_broadcast", (3, 8), (3, 1)), ("outer", (3, 1), (1, 8)),
                  ("rank_mismatch_vector", (3, 8), (8,)), ("scalar", (3, 8), (1,)), ("mismatch", (3, 8), (3, 7)), ("rank3", (2, 3, 8), (1, 1, 8))):
    for dt in ("complex64", "float32"):
        case(f"multiply_{dt}_{tag}", "multiply", {}, {"a": tensor(shape=a, dtype=dt, sample=len(a) - 1), "b": tensor(shape=b, dtype=dt, sample=len(b) - 1)},
             "product", M)
case("multiply_mixed_types", "multiply", {}, {"a": tensor(shape=(8,), sample=0), "b": tensor(shape=(8,), dtype="float32", sample=0)}, "product", M)
case("multiply_strided_operand", "multiply", {}, {"a": tensor(shape=(3, 8), sample=1, batch=0),
                                                  "b": tensor(shape=(8, 2), views=[("select", 1, 1), ("expand", 0)], sample=1)}, "product", M)

# ---- Range, Invert, Window, MultiplyConstant ----------------------------------------------------------------------------------------
for k, (lo, hi) in enumerate(((-100.0, 0.0), (0.0, 1.0), (-1.0, 1.0), (5.0, 5.0), (10.0, -10.0))):
    for tag, kw in layouts(16)[:2] + layouts(16)[6:8]:
        case(f"range_{k}_{tag}", "range", {"min": lo, "max": hi}, {"signal": tensor(dtype="float32", **kw)}, "signal",
             D + "core/range/module_tests.cc (min / max incl. degenerate and inverted, layouts)")
for tag, kw in layouts(16):
    case(f"invert_{tag}", "invert", {}, {"signal": tensor(**kw)}, "signal", D + "dsp/invert/module_tests.cc (axis roles and layouts)")
case("invert_odd_length", "invert", {}, {"signal": tensor(shape=(15,), sample=0)}, "signal", D + "dsp/invert/module_tests.cc (odd length)")
case("invert_f32", "invert", {}, {"signal": tensor(shape=(16,), dtype="float32", sample=0)}, "signal", D + "dsp/invert/module_tests.cc (types)")
for n in (1, 2, 64, 1000):
    case(f"window_{n}", "window", {"size": n}, {}, "window", D + "dsp/window/module_tests.cc:15-53")
case("window_zero", "window", {"size": 0}, {}, "window", D + "dsp/window/module_tests.cc (validation)")
for k, v in enumerate((1.0, -0.5, 1.0 / 4096, 0.0)):
    for dt in ("complex64", "float32"):
        case(f"multiply_constant_{dt}_{k}", "multiply_constant", {"constant": v}, {"factor": tensor(shape=(3, 16), dtype=dt, sample=1, batch=0)}, "product",
             D + "core/multiply_constant/module_tests.cc")

# ---- Pad / Unpad / Fold / Arithmetic / PhaseCorrection / OverlapAdd / AGC / AM ------------------------------------------------------
for tag, kw in layouts(12)[:5]:
    for axis in (-1,):
        case(f"pad_{tag}", "pad", {"size": 5, "axis": axis}, {"unpadded": tensor(**kw)}, "padded", D + "core/pad/module_tests.cc")
        case(f"unpad_{tag}", "unpad", {"size": 5, "axis": axis}, {"padded": tensor(**kw)}, "unpadded", D + "core/unpad/module_tests.cc")
case("pad_axis0_of_rank2", "pad", {"size": 3, "axis": 0}, {"unpadded": tensor(shape=(4, 6), sample=1, batch=0)}, "padded", D + "core/pad/module_tests.cc (axis)")
case("unpad_too_large", "unpad", {"size": 12, "axis": -1}, {"padded": tensor(shape=(12,), sample=0)}, "unpadded", D + "core/unpad/module_tests.cc (validation)")
F = D + "dsp/fold/module_tests.cc:50-391 (uniform / ramp / offset / heads)"
for size, off in ((4, 0), (4, 3), (8, 5), (16, 0), (32, 31)):
    case(f"fold_{size}_{off}", "fold", {"size": size, "offset": off}, {"buffer": tensor(shape=(3, 32), fill="ramp", sample=1, batch=0)}, "buffer", F)
case("fold_heads", "fold", {"size": 8, "offset": 2}, {"buffer": tensor(shape=(2, 3, 32), sample=2, batch=0, channel=1)}, "buffer", F)
case("fold_not_a_divisor", "fold", {"size": 5, "offset": 0}, {"buffer": tensor(shape=(32,), sample=0)}, "buffer", F)
for op in ("add",):
    for axis, sq in ((1, False), (1, True), (0, False), (-1, True)):
        case(f"arithmetic_{op}_axis{axis}_{'squeeze' if sq else 'keep'}", "arithmetic", {"operation": op, "axis": axis, "squeeze": sq},
             {"buffer": tensor(shape=(3, 4, 5), sample=2, batch=0)}, "buffer", D + "core/arithmetic/module_tests.cc")
for k, inc in enumerate((0.0, 0.1, -2.5, 3.141592653589793)):
    case(f"phase_correction_{k}", "phase_correction", {"phaseIncrement": inc}, {"signal": tensor(shape=(2, 3, 16), sample=2, batch=1, channel=0)}, "signal",
         D + "dsp/phase_correction/module_tests.cc (state across three submissions)", cycles=3)
case("overlap_add_batched", "overlap_add", {}, {"buffer": tensor(shape=(4, 16), sample=1, batch=0), "overlap": tensor(shape=(4, 5), sample=1, batch=0)}, "buffer",
     D + "dsp/overlap_add/module_tests.cc:46-236,437 (state across submissions)", cycles=3)
case("overlap_add_heads", "overlap_add", {}, {"buffer": tensor(shape=(2, 3, 16), sample=2, batch=0, channel=1), "overlap": tensor(shape=(2, 3, 5), sample=2, batch=0, channel=1)},
     "buffer", D + "dsp/overlap_add/module_tests.cc:46-236", cycles=2)
for dt in ("complex64", "float32"):
    case(f"agc_{dt}", "agc", {"tileSize": 16, "reference": 0.5}, {"signal": tensor(shape=(3, 64), dtype=dt, sample=1, batch=0)}, "signal",
         D + "dsp/agc/module_tests.cc (tiles, gain limits, two submissions)", cycles=2)
case("agc_tile_does_not_divide", "agc", {"tileSize": 7}, {"signal": tensor(shape=(64,), sample=0)}, "signal", D + "dsp/agc/module_tests.cc (validation)")
case("am_two_submissions", "am", {"sampleRate": 240e3, "dcAlpha": 0.995}, {"signal": tensor(shape=(2, 64), sample=1, batch=0)}, "signal",
     D + "dsp/am/module_tests.cc", cycles=2)
case("add_same_shape", "add", {}, {"a": tensor(shape=(3, 8), sample=1, batch=0), "b": tensor(shape=(3, 8), sample=1, batch=0)}, "sum", D + "core/add/module_tests.cc")
case("add_broadcast", "add", {}, {"a": tensor(shape=(3, 8), sample=1, batch=0), "b": tensor(shape=(1, 8), sample=1)}, "sum", D + "core/add/module_tests.cc")

# ---- ranks 5 .. 8: include/jetstream_hip.h takes tensors of up to JST_MAX_RANK = 8 axes; the reference's iterators
# take 16 (tools/automatic_iterator.hh:182), its tests stop at rank 4 (amplitude/module_tests.cc:475-580).  The same view ops on more axes.
case("multiply_rank5_broadcast", "multiply", {}, {"a": tensor(shape=(2, 1, 3, 2, 8), sample=4, batch=0), "b": tensor(shape=(1, 2, 1, 1, 8), sample=4)}, "product",
     D + "core/multiply/module_tests.cc:84-420 (broadcast), at rank 5")
case("multiply_rank8_permuted", "multiply", {}, {"a": tensor(shape=(2, 1, 2, 1, 2, 1, 3, 8), views=[("permute", 0, 1, 4, 3, 2, 5, 6, 7)], sample=7, batch=0),
                                                 "b": tensor(shape=(8,), sample=0)}, "product",
     D + "core/multiply/module_tests.cc:84-420 (permuted operand), at rank 8")
case("amplitude_rank6_stepped", "amplitude", {}, {"signal": tensor(shape=(2, 4, 1, 2, 3, 16), views=[("range", 1, 1, 4, 2)], sample=5, batch=0)}, "signal",
     D + "dsp/amplitude/module_tests.cc:475-580 (rank-4 non-contiguous), at rank 6")
case("range_rank7", "range", {"min": -2.0, "max": 2.0}, {"signal": tensor(shape=(2, 1, 2, 1, 2, 3, 8), dtype="float32", sample=6, batch=0)}, "signal",
     D + "core/range/module_tests.cc, at rank 7")
case("invert_rank5_strided", "invert", {}, {"signal": tensor(shape=(2, 2, 3, 8, 2), views=[("select", 4, 1)], sample=3, batch=0)}, "signal",
     D + "dsp/invert/module_tests.cc, at rank 5 -> 4 through a strided select")
case("fft_rank5_sample_axis_inside", "fft", {"forward": True}, {"signal": tensor(shape=(2, 2, 8, 3, 2), sample=2, batch=0)}, "signal",
     D + "dsp/fft/module_tests.cc:445-536 (axis handling), at rank 5 with the transform axis inside")
case("fft_rank8_last_axis", "fft", {"forward": False}, {"signal": tensor(shape=(2, 1, 2, 1, 1, 2, 3, 12), sample=7, batch=0)}, "signal",
     D + "dsp/fft/module_tests.cc:445-536, at rank 8")
case("multiply_constant_rank6", "multiply_constant", {"constant": 0.5}, {"factor": tensor(shape=(2, 1, 2, 2, 3, 8), sample=5, batch=0)}, "product",
     D + "core/multiply_constant/module_tests.cc, at rank 6")
case("cast_rank5_ci16", "cast", {"outputType": "CF32"}, {"buffer": tensor(shape=(2, 2, 1, 3, 8), dtype="CI16", sample=4, batch=0)}, "buffer",
     D + "core/cast/module_tests.cc, at rank 5")


# ---- building inputs ----------------------------------------------------------------------------------------------------------------
_CI = {"CI8": np.int8, "CI16": np.int16, "CU8": np.uint8, "CU16": np.uint16}


def storage(case_name, port, spec):
    rng = np.random.default_rng(zlib.crc32(f"{case_name}/{port}".encode()))
    shape, dt, fill = tuple(spec["shape"]), spec["dtype"], spec["fill"]
    if dt in _CI:
        info = np.iinfo(_CI[dt])
        return rng.integers(info.min, info.max + 1, shape + (2,)).astype(_CI[dt])
    ndt = np.dtype(dt)
    count = int(np.prod(shape)) if shape else 1
    if fill == "zeros":
        return np.zeros(shape, ndt)
    if fill == "ramp":
        base = np.arange(count, dtype=np.float64).reshape(shape) * 0.25 - 3.0
    else:
        base = rng.standard_normal(shape)
    if ndt.kind == "c":
        return (base + 1j * (rng.standard_normal(shape) if fill != "ramp" else -0.5 * base)).astype(ndt)
    if ndt.kind in "iu":
        info = np.iinfo(ndt)
        return rng.integers(info.min, info.max + 1, shape).astype(ndt)
    return base.astype(ndt)


def names():
    return [c["name"] for c in CASES]


def by_name(name):
    return next(c for c in CASES if c["name"] == name)


# ---- the reference ---------------------------------------------------------------------------------------------------------------------
def run_reference(c, device="cpu"):
    """(Result code of Module::create / the first failing compute, outputs per cycle or None, output axes).  device "hip": the
    same calls on the reference's DeviceType::HIP (oracle/_ref/libref_jetstream_devhip.so: inputs allocated on the device, the
    module of (HIP, NATIVE), Runtime(HIP)) -- tests/test_gpu_reference_matrix_device_hip.py."""
    from oracle import ref_jetstream as rj
    with rj.RefModule(c["module"], c["config"], device=device) as m:
        for port, spec in c["inputs"].items():
            x = storage(c["name"], port, spec)
            if spec["dtype"] in _CI:
                m.input_ci(port, x, spec["dtype"])
            else:
                m.input(port, x)
            rank = len(spec["shape"])
            for v in spec["views"]:
                if v[0] == "select":
                    tok = []
                    for a in range(rank):
                        tok += [1, v[2], 0, 1] if a == v[1] else [0, 0, 0, 1]
                    m.input_view(port, "slice", tok)
                    rank -= 1
                elif v[0] == "range":
                    tok = []
                    for a in range(rank):
                        tok += [2, v[2], v[3], v[4]] if a == v[1] else [0, 0, 0, 1]
                    m.input_view(port, "slice", tok)
                elif v[0] == "permute":
                    m.input_view(port, "permute", v[1:])
                elif v[0] == "expand":
                    m.input_view(port, "expand_dims", [v[1]])
                    rank += 1
                else:
                    raise KeyError(v[0])
            for key, val in spec["axes"].items():
                m.input_attr(port, key + "Axis", rj.ATTR_INDEX, val)
        code = m.start()
        if code != 0:
            return code, None, None
        outs = []
        for _ in range(c["cycles"]):
            code = m.compute()
            if code != 0:
                return code, None, None
            outs.append(m.output(c["out"]))
        return 0, outs, m.output_axes(c["out"])


# ---- the product -------------------------------------------------------------------------------------------------------------------------
def run_hip(js, c):
    """(accepted?, outputs per cycle or None, output axes or None)."""
    tensors = {}
    try:
        for port, spec in c["inputs"].items():
            x = storage(c["name"], port, spec)
            t = js.Tensor.from_numpy(x, dtype=spec["dtype"]) if spec["dtype"] in _CI else js.Tensor.from_numpy(x)
            for v in spec["views"]:
                if v[0] == "select":
                    t.slice(v[1], v[2], v[2] + 1).squeeze_dims(v[1])
                elif v[0] == "range":
                    t.slice(v[1], v[2], v[3], v[4])
                elif v[0] == "permute":
                    t.permute(v[1:])
                elif v[0] == "expand":
                    t.expand_dims(v[1])
            if spec["axes"]:
                t.set_axes(**spec["axes"])
            tensors[port] = t
        m = js.Module(c["module"], c["config"], tensors, c["name"])
        rt = js.Runtime([m])
    except js.JetstreamError:
        return False, None, None
    outs = []
    try:
        for _ in range(c["cycles"]):
            rt.compute(1)
            outs.append(m.output(c["out"]).numpy())
        axes = m.output(c["out"]).axes
    finally:
        rt.destroy()
    return True, outs, axes


# ---- frozen results ------------------------------------------------------------------------------------------------------------------------
_CACHE = None


def load():
    global _CACHE
    if _CACHE is None:
        z = np.load(_PATH)
        manifest = json.loads(bytes(z["manifest"]).decode())
        _CACHE = {}
        for name, rec in manifest.items():
            outs = [z[f"{name}/out{k}"] for k in range(rec["cycles"])] if rec["code"] == 0 else None
            _CACHE[name] = {"code": rec["code"], "outs": outs, "axes": rec.get("axes"), "cite": rec["cite"]}
    return _CACHE
