"""Seeded random sweeps of the hot path against the oracle: shapes, lengths and parameters nobody
picked by hand.  Bit-exact everywhere (integer bins, float spectra under the generic provider);
the fast providers are held to BASELINE's 1e-5.  Every case is reproducible from its seed."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal, run_module

pytestmark = pytest.mark.gpu


def _lengths(rng, count, hi):
    """Random transform lengths, biased towards the structured ones (smooth, prime, prime * smooth)."""
    out = set()
    while len(out) < count:
        kind = rng.integers(0, 4)
        if kind == 0:
            n = int(rng.integers(1, hi))
        elif kind == 1:   # 2^a 3^b 5^c 7^d 11^e
            n = int(2 ** rng.integers(0, 9) * 3 ** rng.integers(0, 4) * 5 ** rng.integers(0, 3) *
                    7 ** rng.integers(0, 2) * 11 ** rng.integers(0, 2))
        elif kind == 2:   # small factor times a larger prime: generic radix or Bluestein
            n = int(rng.integers(1, 9) * rng.choice([13, 17, 19, 23, 29, 31, 37, 41, 97, 101, 127, 251, 509]))
        else:
            n = int(2 ** rng.integers(0, 13))
        if 1 <= n < hi:
            out.add(n)
    return sorted(out)


@pytest.mark.parametrize("seed", range(20))
def test_complex_fft_random_lengths_and_layouts(js, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for n in _lengths(rng, 25, 9000):
        b = int(rng.integers(1, 6))
        x = csignal(rng, (b, n))
        fwd = bool(rng.integers(0, 2))
        if rng.integers(0, 3) == 0:   # transform along axis 0 of a [n, b] tensor (strided sample axis)
            lead = np.ascontiguousarray(x.T)
            _, out = run_module(js, "fft", {"forward": fwd},
                                {"signal": js.Tensor.from_numpy(lead, sample=0, batch=1)})
            got = np.ascontiguousarray(out["signal"].T)
        else:
            _, out = run_module(js, "fft", {"forward": fwd},
                                {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
            got = out["signal"]
        assert_bit_equal(got, oracle.fft_c2c(x, fwd), f"c2c n={n} b={b} fwd={fwd}")


@pytest.mark.parametrize("seed", range(12))
def test_real_fft_random_lengths(js, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    for n in _lengths(rng, 20, 5000):
        x = rng.standard_normal((int(rng.integers(1, 5)), n)).astype(np.float32)
        t = lambda: js.Tensor.from_numpy(x, sample=1, batch=0)
        fwd = bool(rng.integers(0, 2))
        _, out = run_module(js, "fft", {"forward": fwd}, {"signal": t()})
        assert_bit_equal(out["signal"], oracle.fft_r2r(x, fwd), f"r2r n={n} fwd={fwd}")
        _, out = run_module(js, "fft", {"forward": True, "complexOutput": True}, {"signal": t()})
        assert_bit_equal(out["signal"], oracle.fft_r2c(x), f"r2c n={n}")


@pytest.mark.parametrize("seed", range(24))
def test_spectrum_chain_random_sizes(js, oracle, seed):
    """Window -> ... -> Range -> Spectrogram, fused and unfused, random N (power-of-two kernels and the
    tiled mixed-radix ones), batch, range and height, three cycles of spectrogram state."""
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.choice([64, 256, 500, 1000, 1024, 2048, 3000, 4096, 6000, 8192, 10000, 16384, 20000]))
    b = int(rng.integers(1, 24))
    h = int(rng.choice([16, 100, 256, 300, 512, 1024]))
    lo = float(rng.uniform(-140, -60))
    hi = float(lo + rng.uniform(20, 120))
    fuse = bool(rng.integers(0, 2))
    src = js.Tensor.create("hip", "CF32", (b, n)).set_axes(batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=lo, range_max=hi)
    spec = js.Module("spectrogram", {"height": h}, {"signal": eng.buffer}, "spectrogram")
    rt = js.Runtime(eng.modules + [spec], graph=bool(rng.integers(0, 2)), fuse=fuse)
    bins = np.zeros((h, n), np.float32)
    for cycle in range(3):
        x = csignal(rng, (b, n), scale=10.0 ** rng.uniform(-4, 1))
        x[rng.integers(0, b), rng.integers(0, n)] = 0   # a silent sample somewhere
        src.copy_from(x)
        rt.compute()
        ref = oracle.spectrum_chain(x, lo, hi)
        assert_bit_equal(eng.buffer.numpy(), ref["range"], f"range n={n} b={b} fuse={fuse} cycle={cycle}")
        oracle.spectrogram(bins, ref["range"], h)
        assert_bit_equal(spec.state("frequencyBins").numpy().reshape(h, n), bins,
                         f"spectrogram n={n} b={b} h={h} cycle={cycle}")
    rt.destroy()


@pytest.mark.parametrize("seed", range(20))
def test_spectrogram_random_inputs(js, oracle, seed):
    """The integer bin rule on arbitrary floats: out-of-range values, NaN, infinities, exact bin edges."""
    rng = np.random.default_rng(4000 + seed)
    b, w = int(rng.integers(1, 300)), int(rng.integers(1, 3000))
    h = int(rng.choice([1, 2, 16, 255, 256, 257, 700, 2048]))
    x = rng.uniform(-0.2, 1.2, (b, w)).astype(np.float32)
    edges = (rng.integers(0, h + 1, (b, w)) / np.float32(h)).astype(np.float32)
    x = np.where(rng.random((b, w)) < 0.2, edges, x)
    x[rng.random((b, w)) < 0.01] = np.nan
    x[rng.random((b, w)) < 0.01] = np.inf
    x[rng.random((b, w)) < 0.01] = -np.inf
    spec = js.Module("spectrogram", {"height": h}, {"signal": js.Tensor.from_numpy(x, batch=0, sample=1)})
    rt = js.Runtime([spec], graph=False)
    bins = np.zeros((h, w), np.float32)
    for _ in range(3):
        rt.compute()
        oracle.spectrogram(bins, x, h)
    assert_bit_equal(spec.state("frequencyBins").numpy().reshape(h, w), bins, f"b={b} w={w} h={h}")
    rt.destroy()


def _filter_case(rng):
    while True:
        r = int(rng.choice([1, 2, 4, 5, 8, 10, 16, 20]))
        taps = int(rng.integers(1, 80)) * 2 + 1
        s = int(rng.integers(taps, 4000))
        if r > 1:
            taps = (taps - 1) // r * r + 1            # (taps - 1) % r == 0
            s = max(taps + r, s) // r * r
            if taps < 3 or taps % 2 == 0 or (taps + s - 1) % r != 0:
                continue
        return r, taps, s


@pytest.mark.parametrize("seed", range(40))
def test_filter_block_random_plans(js, oracle, seed):
    """FFT overlap-add chain bit-exact vs the oracle; provider "fast" within 1e-5 of peak of the same."""
    rng = np.random.default_rng(5000 + seed)
    r, taps, s = _filter_case(rng)
    b = int(rng.integers(1, 5))
    sr = 1.0e6 * r
    bw = 1.0e6 if r > 1 else 0.37e6
    heads = int(rng.integers(1, 3))
    center = [0.0] + [float(rng.choice([-0.2, 0.15, 0.3]) * sr) for _ in range(heads - 1)]
    src = js.Tensor.create("hip", "CF32", (b, s)).set_axes(batch=0, sample=1)
    exact = js.Filter(src, sr, bw, center, taps, heads)
    assert exact.plan["resample"] == (r > 1), (r, taps, s, exact.plan)
    rt = js.Runtime(exact.modules, graph=True, fuse=bool(rng.integers(0, 2)))
    fast = rtf = None
    if heads == 1:
        fast = js.Filter(src, sr, bw, center, taps, heads, provider="fast")
        if fast.direct:
            rtf = js.Runtime(fast.modules, graph=True)
    state = {}
    for cycle in range(3):
        x = csignal(rng, (b, s))
        src.copy_from(x)
        rt.compute()
        ref = oracle.filter_block(x, exact.plan, sr, bw, center, taps, state)
        assert_bit_equal(exact.buffer.numpy(), ref, f"r={r} taps={taps} s={s} b={b} heads={heads} cycle={cycle}")
        if rtf is not None:
            rtf.compute()
            err = np.max(np.abs(fast.buffer.numpy() - ref)) / max(1e-30, np.max(np.abs(ref)))
            assert err <= 1e-5, (r, taps, s, b, cycle, err)
    assert fast is None or fast.direct == (r <= min(32, taps) and taps - 1 <= s and r <= 20)


@pytest.mark.parametrize("seed", range(10))
def test_fast_provider_random_chain_within_tolerance(js, oracle, seed):
    rng = np.random.default_rng(6000 + seed)
    n = int(rng.choice([1024, 4096, 6000, 8192]))
    b = int(rng.integers(1, 20))
    x = csignal(rng, (b, n), scale=10.0 ** rng.uniform(-3, 0))
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-110.0, range_max=-10.0, provider="fast")
    rt = js.Runtime(eng.modules, graph=False, fuse=True)
    rt.compute()
    ref = oracle.spectrum_chain(x, -110.0, -10.0)["range"]
    assert np.max(np.abs(eng.buffer.numpy() - ref)) <= 1e-5   # range output lives in [0, 1]
    rt.destroy()


def _random_view(js, rng, dtype_complex):
    """A random dense storage tensor of rank 1..4 and a random strided/offset view of it (steps 1..3 on
    every axis); returns (device view, the same view as a contiguous numpy array)."""
    rank = int(rng.integers(1, 5))
    shape = [int(rng.integers(1, 9)) for _ in range(rank)]
    shape[-1] = int(rng.integers(1, 70))
    store = csignal(rng, shape) if dtype_complex else rng.standard_normal(shape).astype(np.float32)
    t = js.Tensor.from_numpy(store)
    idx = []
    for ax, n in enumerate(shape):
        step = int(rng.integers(1, 4))
        start = int(rng.integers(0, n))
        stop = int(rng.integers(start + 1, n + 1))
        t.slice(ax, start, stop, step)
        idx.append(slice(start, stop, step))
    return t, np.ascontiguousarray(store[tuple(idx)])


@pytest.mark.parametrize("seed", range(30))
def test_elementwise_modules_on_random_strided_views(js, oracle, seed):
    """multiply / add with broadcast partners, amplitude, range, invert, multiply_constant on random
    strided, offset views of random rank: the mixed-radix index decode of the element-wise kernels
    (AutomaticIterator traversal, include/jetstream/tools/automatic_iterator.hh:108-343)."""
    rng = np.random.default_rng(7000 + seed)
    cplx = bool(rng.integers(0, 2))
    ta, a = _random_view(js, rng, cplx)
    # a broadcast partner: same rank, a random subset of the axes collapsed to 1
    bshape = [n if rng.integers(0, 2) else 1 for n in a.shape]
    b = csignal(rng, bshape) if cplx else rng.standard_normal(bshape).astype(np.float32)
    for kind in ("multiply", "add"):
        ref = oracle.multiply(a, b) if kind == "multiply" else oracle.add(a, b)
        out_port = "product" if kind == "multiply" else "sum"
        _, out = run_module(js, kind, {}, {"a": ta, "b": js.Tensor.from_numpy(b)}, outputs=(out_port,))
        assert_bit_equal(out[out_port], ref, f"{kind} seed={seed} shape={a.shape} b={bshape}")
    last = len(a.shape) - 1
    ta.set_axes(sample=last)
    if cplx:
        _, out = run_module(js, "amplitude", {}, {"signal": ta})
        assert_bit_equal(out["signal"], oracle.amplitude(a, a.shape[-1]), f"amplitude seed={seed}")
    else:
        lo = float(rng.uniform(-3, 0))
        hi = float(lo + rng.uniform(0.5, 4))
        _, out = run_module(js, "range", {"min": lo, "max": hi}, {"signal": ta})
        assert_bit_equal(out["signal"], oracle.range_(a, lo, hi), f"range seed={seed}")
    _, out = run_module(js, "invert", {}, {"signal": ta})
    assert_bit_equal(out["signal"], oracle.invert(a, axis=last), f"invert seed={seed}")
