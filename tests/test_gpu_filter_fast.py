"""Provider "fast" of the Filter block: ONE direct-form polyphase FIR + decimate kernel
(csrc/kernels/fir.hip) instead of the FFT overlap-add chain.  Checked against the oracle's
restatement of the reference chain (filter/block_impl.cc:350-582) to the reference's own tolerance
for this block, 1e-5 relative to the peak (filter_engine/block_tests.cc:55-61), over several cycles
(the history tensor carries T-1 samples across rows and submissions), and against a float64
convolution of the whole stream."""
import numpy as np
import pytest

from util import csignal

pytestmark = pytest.mark.gpu

TOL = 1e-5   # relative to the peak of the expected output


def peak_err(got, ref):
    return float(np.max(np.abs(got.astype(np.complex128) - ref.astype(np.complex128))) /
                 max(1e-30, float(np.max(np.abs(ref)))))


@pytest.mark.parametrize("case", [
    dict(sr=20e6, bw=2e6, taps=251, s=1750, b=3),      # /10, J=2 tiles, rows shorter than a tile
    dict(sr=20e6, bw=2e6, taps=251, s=15750, b=4),     # /10, several tiles per row + ragged last tile
    dict(sr=2e6, bw=0.7e6, taps=65, s=960, b=2),       # no resampling (ratio not an integer): r = 1
    dict(sr=8e6, bw=2e6, taps=33, s=4096, b=2),        # /4, J=4
    dict(sr=6e6, bw=2e6, taps=7, s=30, b=5),           # /3, tiny rows: T-1 = 6 history samples
    dict(sr=20e6, bw=2e6, taps=251, s=4000, b=None),   # rank-1 input
    dict(sr=20e6, bw=2e6, taps=101, s=900, b=2, heads=2),
    # round 6, the MFMA form (fir_mfma_kernel: /10, one head, >= 4096 outputs per cycle): both instantiated tap-step counts,
    # totals that are no multiple of a wavefront's 128 outputs, more wave tiles than wavefronts, rows that are no multiple of 16
    dict(sr=20e6, bw=2e6, taps=51, s=40000, b=7),      # fir_mfma_kernel<10, 51>: 28000 outputs, 219 wave tiles
    dict(sr=20e6, bw=2e6, taps=241, s=11110, b=9),     # <10, 101> with fewer taps than it holds; rows of 1111 outputs
    dict(sr=20e6, bw=2e6, taps=251, s=159750, b=12),   # 191700 outputs: 1498 wave tiles on 1020 wavefronts (two rounds)
])
def test_fast_filter_matches_reference_chain(js, oracle, case):
    rng = np.random.default_rng(77)
    sr, bw, taps, s, b = (case[k] for k in ("sr", "bw", "taps", "s", "b"))
    heads = case.get("heads", 1)
    center = [0.0] * heads
    shape = (s,) if b is None else (b, s)
    src = js.Tensor.create("hip", "CF32", shape)
    src.set_axes(sample=0) if b is None else src.set_axes(batch=0, sample=1)
    blk = js.Filter(src, sr, bw, center, taps, heads, provider="fast")
    assert blk.direct and [m.type for m in blk.modules] == ["filter_taps", "fir_taps", "cast", "fir_decimate"]
    plan = blk.plan
    rt = js.Runtime(blk.modules, graph=True)
    state = {}
    xs, ys = [], []
    for cycle in range(3):
        x = csignal(rng, shape)
        src.copy_from(x)
        rt.compute()
        ref = oracle.filter_block(x.reshape(-1, s), plan, sr, bw, center, taps, state)
        got = blk.buffer.numpy()
        if b is None:
            ref = ref[0]
        assert got.shape == ref.shape
        assert peak_err(got, ref) <= TOL, (cycle, peak_err(got, ref))
        xs.append(x.reshape(-1))
        ys.append(got.reshape(-1, heads, got.shape[-1]))
    axes = {"sample": 1, "batch": None, "channel": 0} if b is None else {"sample": 2, "batch": 0, "channel": 1}
    assert blk.buffer.axes == axes
    # the physics, independently: float64 convolution of the whole stream, then every r-th sample
    r = int(np.float32(sr) / np.float32(bw)) if plan["resample"] else 1
    stream = np.concatenate(xs).astype(np.complex128)
    h = oracle.filter_taps(float(np.float32(sr)), float(np.float32(bw)), [0.0], taps)[0].astype(np.complex128)
    full = np.convolve(stream, h)[: stream.size][::r]
    for head in range(heads):
        got = np.concatenate([y[:, head, :] for y in ys], axis=0).reshape(-1)
        assert peak_err(got, full) <= TOL


def test_mfma_form_against_the_direct_form(js, switch):
    """The same provider-fast Filter on the matrix cores (default where the shape allows) and on the vector FMAs
    (JST_FIR_DIRECT): the same sums up to the order of the additions, over three cycles of carried history -- and the
    outputs whose taps reach into the previous cycle come from the same direct-form tile in both."""
    rng = np.random.default_rng(5)
    b, s, taps = 6, 25600, 251
    xs = [csignal(rng, (b, s)) for _ in range(3)]
    outs = {}
    for form in ("direct", "mfma"):
        switch("JST_FIR_DIRECT", "1" if form == "direct" else None)
        src = js.Tensor.create("hip", "CF32", (b, s)).set_axes(batch=0, sample=1)
        blk = js.Filter(src, 20e6, 2e6, [0.0], taps, 1, provider="fast")
        rt = js.Runtime(blk.modules, graph=True)
        got = []
        for x in xs:
            src.copy_from(x)
            rt.compute()
            got.append(blk.buffer.numpy().copy())
        outs[form] = got
        rt.destroy()
    for c, (d, m) in enumerate(zip(outs["direct"], outs["mfma"])):
        assert peak_err(m, d) <= 2e-6, (c, peak_err(m, d))
        assert np.array_equal(m.reshape(-1)[:26].view(np.uint32), d.reshape(-1)[:26].view(np.uint32)), c   # the fix-up tile's outputs
    assert not np.array_equal(outs["direct"][1].view(np.uint32), outs["mfma"][1].view(np.uint32))   # two different kernels did run


def test_fast_provider_keeps_the_fft_chain_when_a_head_is_off_centre(js):
    src = js.Tensor.create("hip", "CF32", (2, 900)).set_axes(batch=0, sample=1)
    blk = js.Filter(src, 20e6, 2e6, [0.0, 3.0e6], 101, 2, provider="fast")
    assert not blk.direct and any(m.type == "overlap_add" for m in blk.modules)
    taps = js.Module("fir_taps", {"decimation": 7}, {"coeffs": js.Tensor.create("hip", "CF32", (1, 101))},
                     provider="fast")
    with pytest.raises(js.JetstreamError, match="Unsupported plan"):      # 900 samples per row, /7
        js.Module("fir_decimate", {}, {"signal": src, "table": taps.output("table")}, provider="fast")
    with pytest.raises(js.JetstreamError, match="must come from a fir_taps module"):
        js.Module("fir_decimate", {}, {"signal": src, "table": js.Tensor.create("hip", "F32", (64,))},
                  provider="fast")
