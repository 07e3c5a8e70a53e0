"""flatten / permutation / signal_axes / ones_tensor / am on the HIP device against numpy views and the
oracle; cases follow the reference's module tests (core/flatten/module_tests.cc:11-220,
core/permutation/module_tests.cc:67-310, core/signal_axes/module_tests.cc:38-300,
core/ones_tensor/module_tests.cc:63-200, dsp/am/module_tests.cc:52-400)."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def dense(js, t):
    """Strided views come home through `duplicate` (the copy the blocks' `contiguous` option inserts)."""
    d = js.Module("duplicate", {}, {"buffer": t})
    rt = js.Runtime([d])
    rt.compute()
    out = d.output("buffer").numpy()
    rt.destroy()
    return out


# ---- flatten ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,complex_", [((4, 8), False), ((2, 3, 5), False), ((3, 7), True)])
def test_flatten_shares_storage_and_clears_roles(js, shape, complex_):
    rng = np.random.default_rng(1)
    x = csignal(rng, shape) if complex_ else rng.standard_normal(shape).astype(np.float32)
    t = js.Tensor.from_numpy(x).set_axes(sample=len(shape) - 1, batch=0)
    m = js.Module("flatten", {}, {"buffer": t})
    out = m.output("buffer")
    assert out.shape == (x.size,) and out.data_ptr == t.data_ptr
    assert out.axes == {"sample": None, "batch": None, "channel": None}  # collapsed geometry: no roles
    assert_bit_equal(out.numpy(), x.reshape(-1), "flatten")


def test_flatten_keeps_roles_of_rank_one_and_rejects_strided_input(js):
    t = js.Tensor.from_numpy(np.arange(8, dtype=np.float32)).set_axes(sample=0)
    assert js.Module("flatten", {}, {"buffer": t}).output("buffer").axes["sample"] == 0
    strided = js.Tensor.from_numpy(np.zeros((4, 8), np.float32)).permute((1, 0))
    with pytest.raises(js.JetstreamError, match="ontiguous"):
        js.Module("flatten", {}, {"buffer": strided})
    bad = js.Tensor.from_numpy(np.zeros((4, 8), np.float32)).set_axes(sample=1, batch=1)
    with pytest.raises(js.JetstreamError):  # malformed metadata (two roles on one axis)
        js.Module("flatten", {}, {"buffer": bad})


# ---- permutation -----------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,perm,complex_", [((2, 3), (1, 0), False), ((2, 3, 4), (2, 0, 1), True),
                                                ((3, 4), (0, 1), False), ((2, 3, 4, 5), (3, 1, 0, 2), False)])
def test_permutation_is_a_strided_view(js, shape, perm, complex_):
    rng = np.random.default_rng(2)
    x = csignal(rng, shape) if complex_ else rng.standard_normal(shape).astype(np.float32)
    t = js.Tensor.from_numpy(x)
    m = js.Module("permutation", {"permutation": list(perm)}, {"buffer": t})
    out = m.output("buffer")
    assert out.shape == tuple(shape[p] for p in perm) and out.data_ptr == t.data_ptr
    assert_bit_equal(dense(js, out), np.ascontiguousarray(np.transpose(x, perm)), "permutation")


def test_permutation_remaps_signal_axes_and_validates(js):
    t = js.Tensor.from_numpy(np.zeros((2, 3, 4), np.float32)).set_axes(batch=0, channel=1, sample=2)
    out = js.Module("permutation", {"permutation": [2, 0, 1]}, {"buffer": t}).output("buffer")
    assert out.axes == {"sample": 0, "batch": 1, "channel": 2}
    for perm, text in (([], "cannot be empty"), ([0, 3, 1], "out of range"), ([0, 1, 1], "more than once"),
                       ([1, 0], "does not match permutation size")):
        with pytest.raises(js.JetstreamError, match=text):
            js.Module("permutation", {"permutation": perm}, {"buffer": t})
    bad = js.Tensor.from_numpy(np.zeros((2, 3), np.float32)).set_axes(sample=5)
    with pytest.raises(js.JetstreamError):
        js.Module("permutation", {"permutation": [1, 0]}, {"buffer": bad})


# ---- signal_axes -----------------------------------------------------------------------------------
def roles(t):
    a = t.axes
    return a["sample"], a["batch"], a["channel"]


def test_signal_axes_layouts(js):
    """core/signal_axes/module_tests.cc:38-220 (sample, batch, channel) per layout string."""
    def run(t, axes):
        out = js.Module("signal_axes", {"axes": axes}, {"buffer": t}).output("buffer")
        assert out.data_ptr == t.data_ptr and out.shape == t.shape
        return roles(out)

    plain = js.Tensor.from_numpy(np.zeros((2, 8), np.float32))
    assert run(plain, "[B, S]") == (1, 0, None)
    assert run(plain, " [ C , S ] ") == (1, None, 0)
    assert run(plain, "[S]") == (0, None, None)  # fewer entries than axes: the rest is unlabelled
    tagged = js.Tensor.from_numpy(np.zeros((2, 8), np.float32)).set_axes(sample=1, batch=0)
    assert run(tagged, "") == (1, 0, None)        # unspecified: metadata passes through
    assert run(tagged, "[*, C]") == (None, 0, 1)  # axis 0 keeps batch, axis 1 becomes channel, sample is dropped
    assert run(tagged, "[*, *]") == (1, 0, None)
    assert run(tagged, "[_, _]") == (None, None, None)
    rank1 = js.Tensor.from_numpy(np.zeros(8, np.float32))
    assert run(rank1, "") == (None, None, None)
    assert run(rank1, "[*]") == (None, None, None)  # the implicit rank-1 sample role is not an attribute
    assert run(rank1, "[_]") == (None, None, None)
    strided = js.Tensor.from_numpy(np.zeros((4, 8), np.float32)).permute((1, 0))
    assert run(strided, "[S, B]") == (0, 1, None)  # DISCONTIGUOUS: views are accepted as they are


@pytest.mark.parametrize("axes", ["B, S", "[]", "[B,]", "[B, X]", "[B, S, C]", "[S, S]", "[BS, C]", "[S, *]"])
def test_signal_axes_rejects_bad_layouts(js, axes):
    t = js.Tensor.from_numpy(np.zeros((2, 8), np.float32)).set_axes(sample=1, batch=0)
    with pytest.raises(js.JetstreamError):
        js.Module("signal_axes", {"axes": axes}, {"buffer": t})


# ---- ones_tensor -----------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,np_dtype", [("F32", np.float32), ("CF32", np.complex64), ("F64", np.float64),
                                            ("CF64", np.complex128)])
def test_ones_tensor(js, dtype, np_dtype):
    m = js.Module("ones_tensor", {"shape": [3, 5, 7], "dataType": dtype})
    out = m.output("buffer")
    assert out.shape == (3, 5, 7) and out.dtype == dtype
    assert np.array_equal(out.numpy(), np.ones((3, 5, 7), np_dtype))  # filled by create()
    out.copy_from(np.zeros((3, 5, 7), np_dtype))
    rt = js.Runtime([m])
    rt.compute(2)  # every compute re-materialises the constant (ones_tensor/module_tests.cc:79-136)
    assert np.array_equal(out.numpy(), np.ones((3, 5, 7), np_dtype))
    rt.destroy()


def test_ones_tensor_feeds_a_multiply_and_validates(js):
    rng = np.random.default_rng(3)
    x = csignal(rng, (4, 64))
    ones = js.Module("ones_tensor", {"shape": [64], "dataType": "CF32"})
    mul = js.Module("multiply", {}, {"a": js.Tensor.from_numpy(x), "b": ones.output("buffer")})
    rt = js.Runtime([ones, mul], graph=True)
    rt.compute(3)
    assert_bit_equal(mul.output("product").numpy(), x, "x * ones")
    rt.destroy()
    assert js.Module("ones_tensor", {}).output("buffer").shape == (1,)  # defaults: F32 [1]
    for cfg, text in (({"shape": []}, "cannot be empty"), ({"shape": [4, 0]}, "cannot be zero"),
                      ({"shape": [4], "dataType": "I32"}, "Invalid data type"),
                      ({"shape": [1 << 40, 1 << 40]}, "layout range"),
                      ({"shape": [1 << 62, 2], "dataType": "F32"}, "byte range"),
                      ({"shape": [1 << 40], "dataType": "CF64"}, "too large")):
        with pytest.raises(js.JetstreamError, match=text):
            js.Module("ones_tensor", cfg)


# ---- am --------------------------------------------------------------------------------------------
def am_signal(rng, n, sr):
    t = np.arange(n) / sr
    env = 1.0 + 0.5 * np.cos(2 * np.pi * 1e3 * t)
    x = env * np.exp(2j * np.pi * 10e3 * t) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


@pytest.mark.parametrize("layout", ["lanes_inner", "lanes_outer"])
def test_am_against_oracle_across_submissions(js, oracle, layout):
    """Bit-exact: lanes are independent, batches of a lane are one sequence, the DC blocker state crosses
    submissions (am/module_tests.cc:259-365); chunk boundaries of the kernel (2048) fall inside."""
    rng = np.random.default_rng(4)
    sr, lanes, batches, samples, alpha = 240e3, 3, 2, 2500, 0.995
    if layout == "lanes_inner":   # [batch, sample, channel]: sample stride = lanes
        shape, axes, lane_ax = (batches, samples, lanes), dict(batch=0, sample=1, channel=2), 2
    else:                         # [channel, batch, sample]
        shape, axes, lane_ax = (lanes, batches, samples), dict(channel=0, batch=1, sample=2), 0
    t = js.Tensor.create("hip", "CF32", shape).set_axes(**axes)
    m = js.Module("am", {"sampleRate": sr, "dcAlpha": alpha}, {"signal": t})
    out = m.output("signal")
    assert out.dtype == "F32" and out.shape == shape and roles(out) == roles(t)
    rt = js.Runtime([m], graph=True)
    refs = [oracle.AmLane(alpha) for _ in range(lanes)]
    for cycle in range(3):
        per_lane = [am_signal(rng, batches * samples, sr).reshape(batches, samples) for _ in range(lanes)]
        if cycle == 1:
            per_lane[1][0, 7] = complex(np.inf, 1.0)  # inf - inf -> NaN poisons that lane for good, like the CPU loop
        x = np.stack(per_lane, axis=lane_ax if lane_ax == 0 else 2)
        if lane_ax == 0:
            x = x.reshape(shape)
        t.copy_from(x)
        rt.compute()
        got = out.numpy()
        for lane in range(lanes):
            ref = refs[lane](per_lane[lane])
            g = np.take(got, lane, axis=lane_ax).reshape(-1)
            nan = np.isnan(ref)  # inf - inf: x86 and gfx950 differ in the sign bit of the default NaN only
            assert np.array_equal(np.isnan(g), nan), (cycle, lane)
            assert_bit_equal(g[~nan], ref[~nan], f"am cycle {cycle} lane {lane}")
            assert nan.any() == (lane == 1 and cycle >= 1)
    rt.destroy()


def test_am_reference_kats_and_validation(js):
    sr, n = 240e3, 1024
    const = js.Tensor.from_numpy(np.ones(n, np.complex64)).set_axes(sample=0)
    m = js.Module("am", {"sampleRate": sr, "dcAlpha": 0.995}, {"signal": const})
    rt = js.Runtime([m])
    rt.compute()
    y = m.output("signal").numpy()
    assert y[0] == 1.0 and abs(y[-1]) < 0.1        # step response of the DC blocker decays (module_tests.cc:52-92)
    assert m.output("signal").attribute("frequency") == 0.0
    rt.destroy()
    tone = js.Tensor.from_numpy(am_signal(np.random.default_rng(5), 2048, sr))  # rank 1: implicit sample axis
    m = js.Module("am", {"sampleRate": sr}, {"signal": tone})
    rt = js.Runtime([m])
    rt.compute()
    y = m.output("signal").numpy()
    assert y.max() - y.min() > 0.01                 # module_tests.cc:130-187
    rt.destroy()
    for cfg, text in (({"sampleRate": 0.0}, "Sample rate"), ({"sampleRate": float("nan")}, "Sample rate"),
                      ({"dcAlpha": 1.0}, "DC alpha"), ({"dcAlpha": -0.1}, "DC alpha")):
        with pytest.raises(js.JetstreamError, match=text):
            js.Module("am", cfg, {"signal": const})
    with pytest.raises(js.JetstreamError, match="signal axis metadata"):
        js.Module("am", {}, {"signal": js.Tensor.from_numpy(np.zeros((2, 8), np.complex64))})
    with pytest.raises(js.JetstreamError, match="complex"):
        js.Module("am", {}, {"signal": js.Tensor.from_numpy(np.zeros(8, np.float32))})


def test_am_dc_alpha_reconfigures_in_place(js, oracle):
    rng = np.random.default_rng(6)
    x = am_signal(rng, 4096, 240e3)
    t = js.Tensor.from_numpy(x).set_axes(sample=0)
    m = js.Module("am", {"dcAlpha": 0.9}, {"signal": t})
    rt = js.Runtime([m], graph=True)
    ref = oracle.AmLane(0.9)
    rt.compute()
    assert_bit_equal(m.output("signal").numpy(), ref(x), "alpha 0.9")
    assert m.reconfigure({"dcAlpha": 0.5}) == "success"
    ref.alpha = np.float32(0.5)
    rt.compute()
    assert_bit_equal(m.output("signal").numpy(), ref(x), "alpha 0.5, state carried")
    rt.destroy()


FLOWGRAPH = """
version: 2
title: view blocks
graph:
  - name: ones
    module: ones_tensor
    device: cpu
    config: {shape: [4, 16], dataType: CF32}
  - name: perm
    module: permutation
    device: cpu
    config: {permutation: [1, 0], contiguous: true}
    input: {buffer: '${graph.ones.output.buffer}'}
  - name: flat
    module: flatten
    device: cpu
    config: {contiguous: false}
    input: {buffer: '${graph.perm.output.buffer}'}
  - name: roles
    module: signal_axes
    device: cpu
    config: {axes: '[S]'}
    input: {buffer: '${graph.flat.output.buffer}'}
  - name: demod
    module: am
    device: cpu
    config: {sampleRate: 48000, dcAlpha: 0.9}
    input: {signal: '${graph.roles.output.buffer}'}
"""


def test_blocks_through_the_flowgraph_loader(js, oracle):
    from cyberether_amd.flowgraph import Flowgraph
    fg = Flowgraph(FLOWGRAPH)
    assert [p["status"] for p in fg.plan] == ["ok"] * 5
    assert [m.type for m in fg.modules] == ["ones_tensor", "permutation", "duplicate", "flatten", "signal_axes", "am"]
    assert fg.output("perm", "buffer").shape == (16, 4) and fg.output("flat", "buffer").shape == (64,)
    rt = fg.runtime()
    rt.compute(2)
    lane = oracle.AmLane(0.9)
    lane(np.ones(64, np.complex64))
    assert_bit_equal(fg.output("demod", "signal").numpy(), lane(np.ones(64, np.complex64)), "second cycle")
    rt.destroy()
