"""The Filter block (FFT overlap-add FIR with spectral-fold resampling) and the Decimator block
on the GPU: bit-exact against the oracle's composition of the same modules, and against an
independent time-domain convolution within the reference's own tolerance (1e-5 relative to
peak, filter_engine/block_tests.cc:55-61), across submissions (overlap state, phase state)."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def two_tone(rng, b, s, sr, f1, f2):
    t = np.arange(b * s) / sr
    x = np.exp(2j * np.pi * f1 * t) + 0.5 * np.exp(2j * np.pi * f2 * t)
    x += 0.01 * (rng.standard_normal(b * s) + 1j * rng.standard_normal(b * s))
    return x.reshape(b, s).astype(np.complex64)


@pytest.mark.parametrize("case", [
    dict(sr=20e6, bw=2e6, center=[0.0], taps=251, s=1750, b=3),            # resample /10, conv 2000
    dict(sr=20e6, bw=2e6, center=[0.0, 3.0e6, -5.0e6], taps=101, s=900, b=2),  # 3 heads, fold offsets
    dict(sr=2e6, bw=0.7e6, center=[0.1e6], taps=65, s=960, b=2),            # no resampling, conv 1024
    dict(sr=20e6, bw=2e6, center=[0.0], taps=251, s=15750, b=2),            # conv 16000: two tiled kernels
    dict(sr=20e6, bw=2e6, center=[0.0, 3.0e6], taps=251, s=15750, b=2),     # the same with two heads (fold offsets)
    dict(sr=20e6, bw=2e6, center=[-1.0e6, 0.0, 3.0e6, 7.0e6], taps=51, s=7950, b=3),  # four heads, conv 8000: one kernel
])
@pytest.mark.parametrize("fuse", [False, True])
def test_filter_block_matches_oracle_chain(js, oracle, case, fuse):
    rng = np.random.default_rng(1235)
    sr, bw, center, taps, s, b = (case[k] for k in ("sr", "bw", "center", "taps", "s", "b"))
    heads = len(center)
    src = js.Tensor.create("hip", "CF32", (b, s)).set_axes(batch=0, sample=1)
    blk = js.Filter(src, sr, bw, center, taps, heads)
    plan = blk.plan
    assert plan == js.filter_plan(sr, bw, center, taps, heads, s)
    rt = js.Runtime(blk.modules, graph=True, fuse=fuse)
    units = rt.units
    if fuse:  # pad -> fft (-> multiply -> fold, any number of heads) collapse into the tiled transform (mixed-radix sizes)
        conv_is_pow2 = plan["convolutionSize"] & (plan["convolutionSize"] - 1) == 0
        one_unit = plan["resample"] and not conv_is_pow2
        assert any(u.startswith("fft_padded_fold(") for u in units) == one_unit, units
        assert any(u.startswith("fft_padded(") for u in units) == (not conv_is_pow2 and not one_unit), units
        assert any(u.startswith("fold_product(") for u in units) == (plan["resample"] and not one_unit), units
        # ifft -> normalize -> unpad -> overlap_add ride on the inverse transform's last store (no phase correction
        # in between, mixed-radix length)
        n_ifft = plan["resamplerSize"] if plan["resample"] else plan["convolutionSize"]
        mixed = n_ifft & (n_ifft - 1) != 0
        centred = all(c == 0.0 for c in center)
        assert any(u.startswith("ifft_unpad_overlap(") for u in units) == (centred and mixed), units
        # ... and with a frequency-shifted head the phase_correction rides there too (round 5)
        assert any(u.startswith("ifft_phase_unpad_overlap(") for u in units) == (not centred and mixed), units
    else:
        assert not any("(" in u for u in units)
    state = {}
    outs = []
    for cycle in range(3):
        x = two_tone(rng, b, s, sr, 0.3e6, 4.0e6)
        src.copy_from(x)
        rt.compute()
        ref = oracle.filter_block(x, plan, sr, bw, center, taps, state)
        got = blk.buffer.numpy()
        assert_bit_equal(got, ref, f"cycle {cycle}")
        outs.append((x, got))
    assert blk.buffer.axes == {"sample": 2, "batch": 0, "channel": 1}
    # independent check of the physics: linear convolution (+ decimation) of the stream
    if not plan["resample"]:
        stream = np.concatenate([x for x, _ in outs], axis=0).reshape(-1).astype(np.complex128)
        for h in range(heads):
            tapsv = oracle.filter_taps(float(np.float32(sr)), float(np.float32(bw)),
                                       [float(np.float32(center[h]))], taps)[0].astype(np.complex128)
            full = np.convolve(stream, tapsv)[: stream.size]
            got_stream = np.concatenate([g[:, h, :] for _, g in outs], axis=0).reshape(-1)
            assert np.max(np.abs(got_stream - full)) <= 1e-5 * max(1.0, np.max(np.abs(full)))


@pytest.mark.parametrize("ratio", [2, 4, 10])
def test_decimator_block(js, oracle, ratio):
    rng = np.random.default_rng(ratio)
    x = (rng.standard_normal((3, 2, 400)) * 100).astype(np.float32)
    src = js.Tensor.from_numpy(x, batch=0, channel=1, sample=2)
    blk = js.Decimator(src, ratio)
    rt = js.Runtime(blk.modules, graph=True)
    rt.compute(2)
    ref = oracle.arithmetic_add(x.reshape(3, 2, 400 // ratio, ratio), 3).reshape(3, 2, 400 // ratio)
    assert_bit_equal(blk.buffer.numpy(), ref)
    assert blk.buffer.axes == {"sample": 2, "batch": 0, "channel": 1}
    c = csignal(rng, (5, 120))
    blk = js.Decimator(js.Tensor.from_numpy(c, batch=0, sample=1), ratio)
    js.Runtime(blk.modules).compute()
    assert_bit_equal(blk.buffer.numpy(), oracle.arithmetic_add(c.reshape(5, 120 // ratio, ratio), 2).reshape(5, -1))


@pytest.mark.parametrize("n,valid,fold,offset,b", [
    (2000, 1750, 200, 0, 5),        # one kernel (whole transforms per lane), /10
    (2000, 1990, 500, 137, 3),      # /4 with a scalar offset
    (6000, 5000, 6000, 17, 2),      # decimation 1: a pure rotation
    (16000, 15750, 1600, 0, 3),     # two kernels: R1 = 128, 1600 mod 128 = 64 -> alias orbit of two block groups
    (16000, 15000, 4000, 4321, 2),  # 4000 mod 128 = 32 -> orbit of four
    (16000, 15000, 3200, 0, 2),     # 3200 mod 128 = 0 -> aliases in the bin's own block
    (20000, 19000, 2000, 0, 2),     # R1 = 8*4*5 = 160 (not a power of two): 2000 mod 160 = 80 -> orbit of two
    (20000, 19990, 5000, 1234, 2),  # 5000 mod 160 = 40 -> orbit of four, offset
    (20000, 19000, 1000, 7, 3),     # 1000 mod 160 = 40, /20
    (50000, 49000, 5000, 49999, 1), # R1 = 400: 5000 mod 400 = 200 -> orbit of two, offset n - 1
    (160000, 159750, 16000, 0, 2),  # SURVEY C3's transform: R1 = 256, orbit of two
    (160000, 159750, 32000, 77777, 1),
])
@pytest.mark.parametrize("spectrum_first", [True, False])
def test_fft_with_multiply_fold_epilogue(js, oracle, n, valid, fold, offset, b, spectrum_first):
    """pad -> fft -> multiply -> fold as ONE unit (the fold's aliases meet in one workgroup of the tiled transform's
    last kernel) against the oracle's composition of the four modules and against the unfused runtime."""
    rng = np.random.default_rng(n + fold + offset)
    x = csignal(rng, (b, valid))
    h = csignal(rng, (1, n))
    outs = {}
    for fuse in (True, False):
        src = js.Tensor.from_numpy(x, batch=0, sample=1)
        hh = js.Tensor.from_numpy(h, batch=0, sample=1)
        pad = js.Module("pad", {"size": n - valid, "axis": 1}, {"unpadded": src}, "pad")
        fft = js.Module("fft", {"forward": True}, {"signal": pad.output("padded")}, "fft")
        spec = fft.output("signal")
        mul = js.Module("multiply", {}, {"a": spec, "b": hh} if spectrum_first else {"a": hh, "b": spec}, "mul")
        fld = js.Module("fold", {"offset": offset, "size": fold},
                        {"buffer": mul.output("product").set_axes(batch=0, sample=1)}, "fold")
        rt = js.Runtime([pad, fft, mul, fld], graph=False, fuse=fuse)
        assert any(u.startswith("fft_padded_fold(") for u in rt.units) == fuse, rt.units
        rt.compute()
        outs[fuse] = fld.output("buffer").numpy()
    spec_ref = oracle.fft_c2c(oracle.pad(x, n - valid, 1))
    prod = oracle.multiply(spec_ref, h) if spectrum_first else oracle.multiply(h, spec_ref)
    ref = oracle.fold(prod, 1, fold, offset)
    assert_bit_equal(outs[True], ref, "fused vs oracle")
    assert_bit_equal(outs[False], ref, "unfused vs oracle")


@pytest.mark.parametrize("shape,axes,size", [
    ((3, 2, 2000), dict(batch=0, channel=1, sample=2), 250),   # one kernel (whole transforms per lane)
    ((2, 1, 16000), dict(batch=0, channel=1, sample=2), 25),   # two kernels (columns + blocks)
    ((3, 600), dict(channel=0, sample=1), 40),                 # no batch axis: every row adds its own state
    ((1, 6000), dict(batch=0, sample=1), 2999),                # overlap nearly as long as the body
])
def test_inverse_fft_with_unpad_overlap_epilogue(js, oracle, shape, axes, size):
    """fft(inverse) -> multiply_constant -> unpad -> overlap_add as ONE unit (scale and body / tail split on the tiled
    transform's last store, one kernel over the overlap region that also rolls the state) against the oracle's
    composition and the unfused runtime, over cycles (state carried across submissions and batches)."""
    rng = np.random.default_rng(sum(shape) + size)
    n = shape[-1]
    batch_axis = axes.get("batch")
    c = float(np.float32(1.0) / np.float32(n))
    runs = {}
    for fuse in (True, False):
        src = js.Tensor.create("hip", "CF32", shape).set_axes(**axes)
        ifft = js.Module("fft", {"forward": False}, {"signal": src}, "ifft")
        norm = js.Module("multiply_constant", {"constant": c}, {"factor": ifft.output("signal")}, "normalize")
        unp = js.Module("unpad", {"size": size, "axis": len(shape) - 1},
                        {"padded": norm.output("product").set_axes(**axes)}, "unpad")
        ola = js.Module("overlap_add", {}, {"buffer": unp.output("unpadded").set_axes(**axes),
                                            "overlap": unp.output("pad").set_axes(**axes)}, "overlap")
        rt = js.Runtime([ifft, norm, unp, ola], graph=True, fuse=fuse)
        assert any(u.startswith("ifft_unpad_overlap(") for u in rt.units) == fuse, rt.units
        runs[fuse] = (src, ola, rt)
    pshape = list(shape)
    pshape[-1] = size
    if batch_axis is not None:
        pshape[batch_axis] = 1
    prev = np.zeros(pshape, np.complex64)
    for cycle in range(3):
        x = csignal(rng, shape)
        y = oracle.fft_c2c(x, forward=False)
        y = (y.view(np.float32) * np.float32(c)).view(np.complex64)   # complex x real: two products
        body, tail = oracle.unpad(y, size, len(shape) - 1)
        ref, prev = oracle.overlap_add(body, tail, prev, batch_axis)
        for fuse in (True, False):
            src, ola, rt = runs[fuse]
            src.copy_from(x)
            rt.compute()
            assert_bit_equal(ola.output("buffer").numpy(), ref, f"cycle {cycle} fuse={fuse}")
            assert_bit_equal(ola.state("previousOverlap").numpy(), prev, f"state, cycle {cycle} fuse={fuse}")
