"""The small-chain fusions of round 5 (cyberether_amd/csrc/modules/chain_fusions.cc) and the one-launch AGC (kernels/agc.hip):
the spectrum_engine block's chain WITH its AGC (spectrum_engine/block_impl.cc:183-197: multiply -> fft -> agc -> amplitude ->
range, one RMS tile per spectrum) on the lengths the reference's multi-fm.yml uses (8000 = 2^6 5^3: the LDS-tiled kernels).
Fused -- `fft_windowed(multiply + fft)`, the AGC as one launch, `amplitude_range(amplitude + range)` -- the chain must leave
exactly what the module-by-module submission leaves and what the oracle computes (window, pocketfft, F64 AGC, libm-exact
amplitude / range), bit for bit."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def _chain(js, x, fuse, **engine):
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, enable_agc=True, **engine)
    rt = js.Runtime(eng.modules, graph=True, fuse=fuse)
    rt._keep = (src, eng)
    return src, eng, rt


def _oracle(oracle, x):
    n = x.shape[-1]
    st = oracle.spectrum_chain(x)
    level = oracle.agc(st["fft"], axis=-1, tile=n)
    amp = oracle.amplitude(level, n)
    return oracle.range_(amp, -100.0, 0.0), level


@pytest.mark.parametrize("n,b", [(8000, 8), (805, 3), (6000, 1)])
def test_agc_spectrum_chain_fused_equals_unfused_and_the_oracle(js, oracle, n, b):
    rng = np.random.default_rng(n)
    t = np.arange(n)
    x = (np.exp(2j * np.pi * 123.25 * t / n)[None, :] * rng.uniform(0.01, 3.0, (b, 1))
         + 0.05 * (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n)))).astype(np.complex64)
    want, level = _oracle(oracle, x)
    outs = {}
    for fuse in (False, True):
        src, eng, rt = _chain(js, x, fuse)
        units = rt.units
        if fuse:
            assert any(u.startswith("fft_windowed(") for u in units), units
            assert any(u.startswith("amplitude_range(") for u in units), units
        else:
            assert not any("(" in u for u in units), units
        rt.compute(3)
        outs[fuse] = eng.buffer.numpy()
        assert_bit_equal(eng.agc.output("signal").numpy(), level, f"AGC output (fuse={fuse}): one tile per spectrum")
        x2 = (x * np.float32(0.5)).astype(np.complex64)     # graph replay on new data
        src.copy_from(x2)
        rt.compute(2)
        assert_bit_equal(eng.buffer.numpy(), _oracle(oracle, x2)[0], f"second input (fuse={fuse})")
        rt.destroy()
    assert_bit_equal(outs[True], outs[False], "fused vs module by module")
    assert_bit_equal(outs[True], want, "fused vs the oracle")


def test_amplitude_range_pair_on_its_own(js, oracle):
    """amplitude -> range outside any spectrum chain (an F32 and a CF32 input), generic provider: one unit, same bits."""
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((4, 300)) + 1j * rng.standard_normal((4, 300))).astype(np.complex64)
    z[0, :3] = [0, 1e-30, 1e30]
    for x, axes in ((z, dict(sample=1, batch=0)), (np.abs(z).astype(np.float32), dict(sample=1, batch=0))):
        src = js.Tensor.from_numpy(x, **axes)
        amp = js.Module("amplitude", {}, {"signal": src}, "amp")
        rg = js.Module("range", {"min": -80.0, "max": 10.0}, {"signal": amp.output("signal")}, "rng")
        rt = js.Runtime([amp, rg], graph=True, fuse=True)
        assert rt.units == ["amplitude_range(amp+rng)"], rt.units
        rt.compute(2)
        want = oracle.range_(oracle.amplitude(x, x.shape[-1]), -80.0, 10.0)
        assert_bit_equal(rg.output("signal").numpy(), want, f"amplitude + range on {x.dtype}")
        rt.destroy()
