"""Modules either side of the hot path vs the oracle and the reference's
own KATs: integer-format cast, add, slice (view), agc.  Everything here is computed with exactly
rounded operations, so the bar is bit-exact; the one exception is noted at the AGC limit cases
(hypot feeds only a comparison)."""
import json
import os

import numpy as np
import pytest

from test_oracle_kats import _agc_case
from util import assert_bit_equal, csignal, run_module

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


# ---- cast (core/cast/module_tests.cc) ----------------------------------------------------------
@pytest.mark.parametrize("name,npdt", [("I8", np.int8), ("U8", np.uint8), ("I16", np.int16),
                                       ("U16", np.uint16), ("I32", np.int32), ("U32", np.uint32)])
def test_cast_real_integers(js, oracle, name, npdt):
    rng = np.random.default_rng(3)
    info = np.iinfo(npdt)
    x = rng.integers(info.min, info.max, size=(5, 333), endpoint=True, dtype=npdt)
    x[0, :4] = [info.min, info.max, 0, 1]
    _, out = run_module(js, "cast", {"outputType": "F32"}, {"buffer": js.Tensor.from_numpy(x)},
                        outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.cast(x), name)


@pytest.mark.parametrize("name,npdt", [("CI8", np.int8), ("CU8", np.uint8), ("CI16", np.int16),
                                       ("CU16", np.uint16), ("CI32", np.int32), ("CU32", np.uint32)])
def test_cast_complex_integers(js, oracle, name, npdt):
    rng = np.random.default_rng(4)
    info = np.iinfo(npdt)
    x = rng.integers(info.min, info.max, size=(3, 1000, 2), endpoint=True, dtype=npdt)
    t = js.Tensor.from_numpy(x, dtype=name, batch=0, sample=1)
    assert t.dtype == name and tuple(t.shape) == (3, 1000)
    m, out = run_module(js, "cast", {"outputType": "CF32"}, {"buffer": t}, outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.cast(x, complex_pairs=True), name)
    assert m.output("buffer").axes == {"sample": 1, "batch": 0, "channel": None}  # attributes propagate


def test_cast_f32_to_cf32_bypass_and_errors(js, oracle):
    rng = np.random.default_rng(5)
    f = rng.standard_normal((4, 64)).astype(np.float32)
    _, out = run_module(js, "cast", {"outputType": "CF32"}, {"buffer": js.Tensor.from_numpy(f)},
                        outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.cast(f))
    # same dtype: the output aliases the input (cast/module_impl.cc:96-100)
    t = js.Tensor.from_numpy(f)
    m = js.Module("cast", {"outputType": "F32"}, {"buffer": t})
    assert m.output("buffer").data_ptr == t.data_ptr
    # strided input view (DISCONTIGUOUS taint)
    x = rng.integers(-128, 127, size=(6, 40), dtype=np.int8)
    tv = js.Tensor.from_numpy(x).slice(1, 4, 36, 2)
    _, out = run_module(js, "cast", {"outputType": "F32"}, {"buffer": tv}, outputs=("buffer",))
    assert_bit_equal(out["buffer"], oracle.cast(np.ascontiguousarray(x[:, 4:36:2])))
    with pytest.raises(js.JetstreamError, match="Invalid output type 'Q7'"):
        js.Module("cast", {"outputType": "Q7"}, {"buffer": js.Tensor.from_numpy(f)})
    with pytest.raises(js.JetstreamError, match="Unsupported conversion 'I8' -> 'CF32'"):
        js.Module("cast", {"outputType": "CF32"}, {"buffer": js.Tensor.from_numpy(x)})


# ---- add (core/add/module_tests.cc) ------------------------------------------------------------
def test_add_broadcast_and_attributes(js, oracle):
    rng = np.random.default_rng(6)
    a = csignal(rng, (7, 130))
    b = csignal(rng, (1, 130))
    ta = js.Tensor.from_numpy(a, batch=0, sample=1).set_attribute("sampleRate", 2.0e6)
    tb = js.Tensor.from_numpy(b, sample=1).set_attribute("sampleRate", 0.0)
    m, out = run_module(js, "add", {}, {"a": ta, "b": tb}, outputs=("sum",))
    assert_bit_equal(out["sum"], oracle.add(a, np.broadcast_to(b, a.shape)))
    assert m.output("sum").axes == {"sample": 1, "batch": 0, "channel": None}
    fa = rng.standard_normal((3, 1, 8)).astype(np.float32)
    fb = rng.standard_normal((5, 1)).astype(np.float32)
    _, out = run_module(js, "add", {}, {"a": js.Tensor.from_numpy(fa), "b": js.Tensor.from_numpy(fb)},
                        outputs=("sum",))
    assert_bit_equal(out["sum"], (fa + fb).astype(np.float32))
    with pytest.raises(js.JetstreamError, match=r"\[MODULE_ADD\] Input shapes .* are not broadcastable"):
        js.Module("add", {}, {"a": js.Tensor.from_numpy(fa), "b": js.Tensor.from_numpy(np.zeros((3, 2, 7), np.float32))})
    with pytest.raises(js.JetstreamError, match="MODULE_ADD_NATIVE_HIP"):
        js.Module("add", {}, {"a": js.Tensor.from_numpy(fa), "b": js.Tensor.from_numpy(a)})


# ---- slice (core/slice/module_tests.cc) --------------------------------------------------------
@pytest.mark.parametrize("text,index", [
    ("[:, 1, :]", np.s_[:, 1, :]),
    ("[0, :]", np.s_[0, :]),
    ("[..., 3]", np.s_[..., 3]),
    ("[1:3, ..., 2:10:3]", np.s_[1:3, ..., 2:10:3]),
    ("[ ]", np.s_[...]),
    ("[2]", np.s_[2]),
    ("[:, :2]", np.s_[:, :2]),
])
def test_slice_views(js, text, index):
    rng = np.random.default_rng(7)
    x = csignal(rng, (4, 3, 16))
    t = js.Tensor.from_numpy(x, batch=0, channel=1, sample=2)
    m = js.Module("slice", {"slice": text}, {"buffer": t})
    view = m.output("buffer")
    ref = x[index]
    assert tuple(view.shape) == ref.shape
    assert view.data_ptr == t.data_ptr  # a view: no copy, no kernel
    # materialise through duplicate (what the slice BLOCK does with contiguous: true)
    _, out = run_module(js, "duplicate", {}, {"buffer": view}, outputs=("buffer",))
    assert_bit_equal(out["buffer"], np.ascontiguousarray(ref))


def test_slice_axes_and_errors(js):
    x = np.zeros((4, 3, 16), np.complex64)
    t = js.Tensor.from_numpy(x, batch=0, channel=1, sample=2)
    v = js.Module("slice", {"slice": "[:, 1, :]"}, {"buffer": t}).output("buffer")
    assert v.axes == {"sample": 1, "batch": 0, "channel": None}  # the indexed role disappears
    v = js.Module("slice", {"slice": "[2, ...]"}, {"buffer": t}).output("buffer")
    assert v.axes == {"sample": 1, "batch": None, "channel": 0}
    for text, msg in (("", "cannot be empty"), ("1, 2", "Missing brackets"), ("[1,,2]", "Empty token"),
                      ("[a]", "Invalid token 'a'"), ("[1:2:3:4]", "Invalid token"), ("[::0]", "step cannot be zero"),
                      ("[..., ...]", "Ellipsis can only appear once"), ("[4]", "out of range"),
                      ("[:, :, :, :]", "exceeds dimensions"), ("[0:5]", "exceeds dimension"), ("[-1]", "Invalid token")):
        with pytest.raises(js.JetstreamError, match=msg):
            js.Module("slice", {"slice": text}, {"buffer": t})


# ---- agc (dsp/agc/module_tests.cc) -------------------------------------------------------------
def _gpu_agc(js):
    cfgmap = {"tile": "tileSize", "reference": "reference", "epsilon": "epsilon", "min_gain": "minGain",
              "max_gain": "maxGain", "max_gain_change": "maxGainChange"}

    def fn(x, axis, **kw):
        axes = {"sample": axis % x.ndim}
        if x.ndim == 2:
            axes["channel"] = 1 - axes["sample"]
        t = js.Tensor.from_numpy(x, **axes)
        _, out = run_module(js, "agc", {cfgmap[k]: v for k, v in kw.items()}, {"signal": t})
        return out["signal"]
    return fn


def test_agc_reference_kats(js):
    for case in KATS["agc"]:
        _agc_case(_gpu_agc(js), case)


@pytest.mark.parametrize("dtype", ["F32", "CF32"])
@pytest.mark.parametrize("shape,axis,tile", [((3, 5000), 1, 1024), ((4100, 6), 0, 1000), ((2, 3, 257), 2, 64),
                                             ((8, 4096), 1, 4096), ((5, 100), 1, 1)])
def test_agc_random_bit_exact(js, oracle, dtype, shape, axis, tile):
    rng = np.random.default_rng(8)
    scale = np.exp(rng.uniform(-6, 6, size=shape)).astype(np.float32)  # exercise the gain limits
    x = csignal(rng, shape) * scale if dtype == "CF32" else (rng.standard_normal(shape) * scale).astype(np.float32)
    x = x.astype(np.complex64 if dtype == "CF32" else np.float32)
    axes = {"sample": axis}
    if len(shape) == 2:
        axes["batch"] = 1 - axis
    else:
        axes.update(batch=0, channel=1)
    t = js.Tensor.from_numpy(x, **axes)
    m, out = run_module(js, "agc", {"tileSize": tile, "maxGainChange": 1.5}, {"signal": t})
    assert_bit_equal(out["signal"], oracle.agc(x, axis, tile=tile, max_gain_change=1.5), f"{dtype} {shape}")
    assert m.output("signal").axes["sample"] == axis


def test_agc_validation(js):
    t = js.Tensor.from_numpy(np.zeros(16, np.float32))
    for cfg, msg in (({"tileSize": 0}, "Tile size"), ({"reference": 0.0}, "Reference"),
                     ({"reference": float("inf")}, "Reference"), ({"epsilon": 0.0}, "Epsilon"),
                     ({"minGain": 0.0}, "Minimum gain"), ({"maxGain": 0.005}, "Maximum gain must"),
                     ({"maxGain": float("nan")}, "Maximum gain must"), ({"maxGainChange": 0.5}, "gain change"),
                     ({"maxGainChange": float("nan")}, "gain change")):
        with pytest.raises(js.JetstreamError, match=msg):
            js.Module("agc", cfg, {"signal": t})
    with pytest.raises(js.JetstreamError, match="Unsupported data type 'U8'"):
        js.Module("agc", {}, {"signal": js.Tensor.from_numpy(np.zeros(16, np.uint8))})
    with pytest.raises(js.JetstreamError, match="valid signal axis metadata"):
        js.Module("agc", {}, {"signal": js.Tensor.from_numpy(np.zeros((4, 4), np.float32))})


def test_spectrum_engine_with_agc(js, oracle):
    """spectrum_engine/block_impl.cc:185-196: agc(tileSize = N) between fft and amplitude."""
    rng = np.random.default_rng(9)
    x = csignal(rng, (6, 1024), scale=0.05)
    t = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(t, enable_agc=True, enable_scale=True, range_min=-80.0, range_max=0.0)
    rt = js.Runtime(eng.modules, graph=True, fuse=True)
    rt.compute(3)
    got = eng.buffer.numpy()
    rt.destroy()
    win = oracle.invert(oracle.window(1024))
    spec = oracle.fft_c2c(oracle.multiply(x, win[None, :]))
    ref = oracle.range_(oracle.amplitude(oracle.agc(spec, 1, tile=1024), 1024), -80.0, 0.0)
    assert_bit_equal(got, ref)
