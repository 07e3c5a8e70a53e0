"""Elementwise HIP modules vs the oracle, bit for bit: window, invert, multiply (broadcast,
strided), amplitude (incl. exact zero -> -inf, amplitude/module_tests.cc:419-473, rank-4
non-contiguous :475-580), range, multiply_constant, and the device restatement of libm tanhf."""
import ctypes as C

import numpy as np
import pytest

from util import assert_bit_equal, csignal, run_module

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 7, 64, 1000, 4096, 65536])
def test_window_bit_exact(js, oracle, n):
    _, out = run_module(js, "window", {"size": n}, {}, outputs=("window",))
    assert_bit_equal(out["window"], oracle.window(n), f"window {n}")


@pytest.mark.parametrize("n", [2, 64, 4096, 5, 63])
def test_invert_even_and_odd(js, oracle, n):
    w = oracle.window(n)
    t = js.Tensor.from_numpy(w, sample=0)
    _, out = run_module(js, "invert", {}, {"signal": t})
    assert_bit_equal(out["signal"], oracle.invert(w), f"invert {n}")
    f = np.linspace(-1, 1, 3 * n, dtype=np.float32).reshape(3, n)
    _, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(f, sample=1, batch=0)})
    assert_bit_equal(out["signal"], oracle.invert(f, axis=1), f"invert f32 {n}")
    g = csignal(np.random.default_rng(n), (n, 4))
    _, out = run_module(js, "invert", {}, {"signal": js.Tensor.from_numpy(g, sample=0, batch=1)})
    assert_bit_equal(out["signal"], oracle.invert(g, axis=0), f"invert leading axis {n}")


def test_multiply_broadcast_and_strided(js, oracle):
    rng = np.random.default_rng(11)
    a = csignal(rng, (5, 3, 64))
    b = csignal(rng, (64,))
    ta = js.Tensor.from_numpy(a, sample=2, batch=0, channel=1)
    tb = js.Tensor.from_numpy(b, sample=0)
    m, out = run_module(js, "multiply", {}, {"a": ta, "b": tb}, outputs=("product",))
    assert_bit_equal(out["product"], oracle.multiply(a, b.reshape(1, 1, 64)))
    assert m.output("product").axes == {"sample": 2, "batch": 0, "channel": 1}
    # column vector times row vector, F32
    c = rng.standard_normal((7, 1)).astype(np.float32)
    d = rng.standard_normal((1, 9)).astype(np.float32)
    _, out = run_module(js, "multiply", {}, {"a": js.Tensor.from_numpy(c), "b": js.Tensor.from_numpy(d)},
                        outputs=("product",))
    assert_bit_equal(out["product"], oracle.multiply(c, d))
    # strided + offset operand
    store = csignal(rng, (8, 128))
    ts = js.Tensor.from_numpy(store)
    ts.slice(0, 1, 8, 3).slice(1, 10, 74, 1).set_axes(sample=1, batch=0)
    _, out = run_module(js, "multiply", {}, {"a": ts, "b": tb}, outputs=("product",))
    assert_bit_equal(out["product"], oracle.multiply(np.ascontiguousarray(store[1:8:3, 10:74]), b.reshape(1, 64)))
    # non-finite operands take the std::complex recovery branch like libgcc __mulsc3
    e = np.array([complex(np.inf, 1), complex(np.nan, np.inf), complex(0, -np.inf), 1 + 1j], np.complex64)
    g = np.array([complex(1, np.nan), complex(2, 2), complex(np.inf, 0), complex(np.nan, np.nan)], np.complex64)
    _, out = run_module(js, "multiply", {}, {"a": js.Tensor.from_numpy(e), "b": js.Tensor.from_numpy(g)},
                        outputs=("product",))
    ref = oracle.multiply(e, g)
    assert np.array_equal(np.isnan(out["product"].view(np.float32)), np.isnan(ref.view(np.float32)))
    fin = ~np.isnan(ref.view(np.float32))
    assert np.array_equal(out["product"].view(np.float32)[fin], ref.view(np.float32)[fin])


def test_amplitude_bit_exact_and_kats(js, oracle):
    rng = np.random.default_rng(12)
    x = csignal(rng, (6, 4096), scale=37.0)
    x[0, :8] = 0
    x[1, 0] = complex(1e-30, 0)          # subnormal magnitudes squared flush to the frexp path
    x[1, 1] = complex(3e19, 3e19)        # re*re overflows -> inf magnitude
    x[2, 0] = complex(-0.0, 0.0)
    _, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    ref = oracle.amplitude(x, 4096)
    assert np.all(np.isneginf(out["signal"][0, :8]))
    assert_bit_equal(out["signal"], ref)
    f = (rng.standard_normal((3, 128)) * 5).astype(np.float32)
    f[0, 0] = 0.0
    _, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(f, sample=1, batch=0)})
    assert_bit_equal(out["signal"], oracle.amplitude(f, 128))
    # channel-only tensor: normalisation size 1 (amplitude/module_impl.cc:29-31)
    ch = np.full(16, 2.0, np.float32)
    _, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(ch, channel=0)})
    assert_bit_equal(out["signal"], oracle.amplitude(ch, 1))
    # reference KAT: constant 1+0j over 64 samples is -36.12 dB within 0.5 dB
    one = np.ones(64, np.complex64)
    _, out = run_module(js, "amplitude", {}, {"signal": js.Tensor.from_numpy(one)})
    assert np.max(np.abs(out["signal"] - 20 * np.log10(1 / 64))) < 0.5


def test_amplitude_rank4_noncontiguous(js, oracle):
    rng = np.random.default_rng(13)
    store = csignal(rng, (3, 4, 5, 32))
    t = js.Tensor.from_numpy(store)
    t.permute((2, 0, 1, 3)).slice(3, 4, 28, 2).set_axes(sample=3, batch=0)
    host = np.ascontiguousarray(store.transpose(2, 0, 1, 3)[..., 4:28:2])
    _, out = run_module(js, "amplitude", {}, {"signal": t})
    assert_bit_equal(out["signal"], oracle.amplitude(host, host.shape[3]))


def test_range_bit_exact_vs_host_libm(js, oracle):
    rng = np.random.default_rng(14)
    x = (rng.standard_normal((4, 4096)) * 60 - 50).astype(np.float32)
    x[0, :4] = [-np.inf, np.inf, np.nan, -0.0]
    _, out = run_module(js, "range", {"min": -100.0, "max": 0.0},
                        {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    ref = oracle.range_(x, -100.0, 0.0)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(out["signal"]), nan)
    assert_bit_equal(out["signal"][~nan], ref[~nan])
    _, out = run_module(js, "range", {"min": 3.0, "max": 3.0}, {"signal": js.Tensor.from_numpy(x[1])})
    assert np.all(out["signal"] == 0.5)  # degenerate range (range/module_impl.cc:56-58)
    _, out = run_module(js, "range", {"min": 0.0, "max": -100.0},
                        {"signal": js.Tensor.from_numpy(x[1])})
    assert_bit_equal(out["signal"], oracle.range_(x[1], 0.0, -100.0))  # swapped bounds


def test_device_tanhf_is_the_hosts_libm_tanhf(js):
    """4M floats spread over the whole binary32 range + a dense sweep of [-12, 12]."""
    import torch

    u = np.arange(0, 2 ** 32, 1021, dtype=np.uint64).astype(np.uint32)
    sweep = np.concatenate([u.view(np.float32), np.linspace(-12, 12, 1 << 20, dtype=np.float32)])
    x = torch.from_numpy(sweep).cuda()
    y = torch.empty_like(x)
    r = js._lib.jst_probe_tanhf(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel())
    assert r == 0, js._lib.jst_last_error()
    got = y.cpu().numpy()
    libm = C.CDLL("libm.so.6")
    libm.tanhf.restype = C.c_float
    libm.tanhf.argtypes = [C.c_float]
    idx = np.random.default_rng(0).choice(sweep.size, 300000, replace=False)
    ref = np.array([libm.tanhf(float(v)) for v in sweep[idx]], np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got[idx]), nan)
    assert_bit_equal(got[idx][~nan], ref[~nan], "tanhf")


def test_multiply_constant(js):
    rng = np.random.default_rng(15)
    x = csignal(rng, (3, 100))
    _, out = run_module(js, "multiply_constant", {"constant": 1.0 / 160000.0},
                        {"factor": js.Tensor.from_numpy(x, sample=1, batch=0)}, outputs=("product",))
    c = np.float32(1.0 / 160000.0)
    ref = (x.real * c + 1j * (x.imag * c)).astype(np.complex64)
    assert_bit_equal(out["product"], ref)
