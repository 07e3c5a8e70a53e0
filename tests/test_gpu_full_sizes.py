"""BASELINE.json's other configurations at FULL size (config 2 lives in test_gpu_chain.py):
bit-exact against the oracle where the oracle finishes in seconds, plus size-independent
properties (replicated batches give replicated results; steady-state overlap-add) on the rest."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal

pytestmark = pytest.mark.gpu


def test_config5_65536_point_fused_chain(js, oracle):
    """Window -> 65536-pt FFT -> Amplitude -> Range -> Lineplot(avg 8) on 16 batches: the LDS-tiled
    two-kernel path with the window / amplitude / range functors fused in."""
    n, b = 65536, 16
    rng = np.random.default_rng(65536)
    x = csignal(rng, (b, n), 0.02)
    x[:, :] += (0.5 * np.exp(2j * np.pi * 0.123 * np.arange(n))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-110.0, range_max=-5.0)
    lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
    rt = js.Runtime(eng.modules + [lp], graph=True, fuse=True)
    assert any(u.startswith("spectrum_fused(") for u in rt.units)
    avg = np.zeros(n, np.float32)
    ref = oracle.spectrum_chain(x, -110.0, -5.0)["range"]
    for _ in range(3):
        rt.compute(1)
        oracle.lineplot(avg, ref, averaging=8)
    assert_bit_equal(eng.buffer.numpy(), ref, "65536-pt fused chain")
    assert_bit_equal(lp.state("averagingBuffer").numpy(), avg, "averaged PSD")
    peak = int(np.argmax(avg))
    assert abs(peak - (n // 2 + round(0.123 * n))) <= 1   # invert = fftshift: tone at n/2 + f*n
    rt.destroy()


@pytest.mark.parametrize("fuse", [False, True])
def test_config3_filter_block_full_size(js, oracle, fuse):
    """251 taps, /10 resampling, S = 159750 (convolution 160000 = 8*8*4*5^4), 100 batches of ONE
    repeated input row: rows 0 and 1 are checked against the oracle (overlap state empty, then
    carried); every later row must equal row 1 bit for bit (steady state of overlap-add)."""
    b, s, taps, sr, bw = 100, 159750, 251, 20e6, 2e6
    rng = np.random.default_rng(3)
    t = np.arange(s) / sr
    row = (np.exp(2j * np.pi * 0.3e6 * t) + 0.5 * np.exp(2j * np.pi * 4.0e6 * t)).astype(np.complex64)
    row += csignal(rng, (s,), 0.01)
    x = np.ascontiguousarray(np.broadcast_to(row, (b, s)))
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    blk = js.Filter(src, sr, bw, [0.0], taps, 1)
    assert blk.plan["convolutionSize"] == 160000 and blk.plan["resamplerSize"] == 16000
    rt = js.Runtime(blk.modules, graph=True, fuse=fuse)
    rt.compute(1)
    got = blk.buffer.numpy()
    assert got.shape == (b, 1, 15975)
    state = {}
    ref = oracle.filter_block(x[:3], blk.plan, sr, bw, [0.0], taps, state)
    assert_bit_equal(got[:3], ref, "first rows vs oracle")
    assert np.array_equal(got[2:].view(np.uint32), np.broadcast_to(got[1:2], got[2:].shape).view(np.uint32))
    # the 0.3 MHz tone is in the 2 MHz passband, the 4 MHz tone is rejected by > 60 dB
    spec = np.abs(np.fft.fft(got[5, 0].astype(np.complex128)))
    k_pass = round(0.3e6 / 2e6 * got.shape[2])
    assert spec[k_pass] > 1000 * np.median(spec)
    rt.destroy()


@pytest.mark.parametrize("fuse", [False, True])
def test_config3_literal_distinct_rows_two_cycles(js, oracle, fuse):
    """SURVEY 8(d) C3 as written: CF32[100, 159750], two tones (0.3 MHz in band, 4 MHz out of band) + complex AWGN
    from default_rng(1235) -- every row DISTINCT -- 251 taps, /10, two compute cycles with a second distinct batch,
    so that the overlap state crosses rows AND the compute boundary at the 160000-point size.  Rows 0..7 and
    92..99 of both cycles are compared bit for bit with the oracle; overlap-add state is sequential along the
    batch axis and depends on the previous row only, so the oracle walks rows 91..99 to hand the carried tail
    to the next cycle's row 0."""
    b, s, taps, sr, bw = 100, 159750, 251, 20e6, 2e6
    rng = np.random.default_rng(1235)
    t = np.arange(s) / sr
    tones = (np.exp(2j * np.pi * 0.3e6 * t) + 0.5 * np.exp(2j * np.pi * 4.0e6 * t)).astype(np.complex64)
    batches = [(tones[None, :] + csignal(rng, (b, s), 0.01)).astype(np.complex64) for _ in range(2)]
    src = js.Tensor.from_numpy(batches[0], batch=0, sample=1)
    blk = js.Filter(src, sr, bw, [0.0], taps, 1)
    assert blk.plan["convolutionSize"] == 160000 and blk.plan["resamplerSize"] == 16000
    rt = js.Runtime(blk.modules, graph=True, fuse=fuse)
    state = {}   # the oracle's carried overlap tail: what the row before the rows under test left behind
    for cycle, x in enumerate(batches):
        if cycle:
            src.copy_from(x)
        rt.compute(1)
        got = blk.buffer.numpy()
        assert got.shape == (b, 1, 15975)
        head = oracle.filter_block(x[:8], blk.plan, sr, bw, [0.0], taps, state)
        assert_bit_equal(got[:8], head, f"cycle {cycle}: rows 0..7 (carried state from the previous cycle's last row)")
        state = {}
        tail = oracle.filter_block(x[91:], blk.plan, sr, bw, [0.0], taps, state)   # row 91 primes the tail
        assert_bit_equal(got[92:], tail[1:], f"cycle {cycle}: rows 92..99")
        assert len({got[r].tobytes() for r in range(0, b, 9)}) == len(range(0, b, 9))   # rows really are distinct
    rt.destroy()


def test_config4_wbfm_chain_full_rate(js, oracle):
    """20 MS/s -> Filter(/100) -> FM (wide, 75 us) -> Decimator(/4): one stereo lane, 10 batches."""
    b, s, taps, sr, bw = 10, 202400, 101, 20e6, 200e3   # conv 202500 = 2^2 * 3^4 * 5^4
    tt = np.arange(b * s) / sr
    audio = 0.45 * np.sin(2 * np.pi * 1e3 * tt) + 0.1 * np.sin(2 * np.pi * 19e3 * tt)
    x = np.exp(2j * np.pi * 75e3 * np.cumsum(audio) / sr).astype(np.complex64).reshape(b, s)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    filt = js.Filter(src, sr, bw, [0.0], taps, 1)
    squeeze = js.Module("squeeze_dims", {"axis": 1}, {"buffer": filt.buffer}, "squeeze_head")
    iq = squeeze.output("buffer").set_axes(batch=0, sample=1)
    fm = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3}, {"signal": iq}, "fm")
    dec = js.Decimator(fm.output("signal"), 4)
    rt = js.Runtime(filt.modules + [squeeze, fm] + dec.modules, graph=True, fuse=True)
    rt.compute(1)
    base = oracle.filter_block(x, filt.plan, sr, bw, [0.0], taps, {})
    assert_bit_equal(filt.buffer.numpy(), base, "channel filter at 20 MS/s")
    lane = oracle.FmLane("wide", "75us", 200e3)
    stereo = np.asarray(lane(np.ascontiguousarray(base[:, 0, :]))).reshape(b, 2024, 2)
    got_fm = fm.output("signal").numpy()
    assert_bit_equal(got_fm, np.asarray(stereo, np.float32), "wide FM decode (libm sinf/cosf/atan2f restated)")
    ref_dec = oracle.arithmetic_add(np.ascontiguousarray(got_fm.reshape(b, 506, 4, 2)), 2).reshape(b, 506, 2)
    assert_bit_equal(dec.buffer.numpy(), ref_dec, "integrate-and-dump /4 of the device FM output")
    rt.destroy()
