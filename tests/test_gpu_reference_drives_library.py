"""GPU: the REFERENCE drives the library.  oracle/_ref/libref_jetstream_hip.so is the reference's own core (registry,
Module::create, native-CPU runtime, synchronous scheduler, blocks -- compiled in place, oracle/ref_jetstream_build.sh) with the
reference-side `provider: mi355x` modules of integration/mi355x_provider/ linked in, and linked against
cyberether_amd/lib/libjetstream_hip.so.  Here the reference's
    Registry::BuildModule(type, CPU, NATIVE, "mi355x")  (include/jetstream/registry.hh:119-125, src/registry.cc:605-618)
    -> Module::create (src/module.cc:47-212) -> Runtime::compute (src/runtime/native/cpu/impl.cc:98-148)
and Flowgraph::blockCreate(spectrum_engine, provider = "mi355x") (src/block_impl.cc:46-53: every module of the block is
built with the block's provider) -> Flowgraph::compute run the HIP kernels, and their outputs are compared BIT FOR BIT with
what the same reference calls produce with provider "generic" (its own CPU modules) on the same inputs -- SURVEY 8(b)'s
drop-in boundary exercised from the reference's side, not from ours."""
import numpy as np
import pytest

from oracle import ref_jetstream as rj
from util import assert_bit_equal

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not rj.hip_library_available(), reason="oracle/_ref/libref_jetstream_hip.so not built")]

MODULES = ("cast", "window", "invert", "reshape", "multiply", "fft", "amplitude", "range", "spectrogram")


@pytest.fixture(scope="module", autouse=True)
def reference_linked_against_the_library(js):  # `js`: the product library is loaded (and the device selected) first
    rj.use_hip_library()
    yield


def cnoise(rng, *shape, scale=1.0):
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * scale).astype(np.complex64)


def c1_tone(oracle, rows=1, n=4096, sigma=1e-3, seed=1234):
    """C1 (row 0: the off-bin CW tone) and C2-style rows (tone at bin 100.25 + r plus AWGN)."""
    rng = np.random.default_rng(seed)
    fs = 2.0e6
    x = np.empty((rows, n), np.complex64)
    for r in range(rows):
        x[r], _ = oracle.signal_cosine(n, 1.0, (100.25 + r) * fs / n, fs)
    if rows > 1:
        x[1:] += cnoise(rng, rows - 1, n, scale=sigma)
    return x


def both(mtype, cfg, inputs, out, cycles=1):
    """The same reference calls with provider "generic" and provider "mi355x"; returns the two outputs."""
    got = []
    for provider in ("generic", "mi355x"):
        with rj.RefModule(mtype, cfg, provider=provider) as m:
            for port, (x, axes) in inputs.items():
                m.input(port, x, **axes)
            assert m.start() == 0, f"{mtype} ({provider}): Module::create failed"
            for _ in range(cycles):
                assert m.compute() == 0, f"{mtype} ({provider}): Runtime::compute failed"
            got.append(m.output(out))
    return got


def test_the_registry_holds_the_provider():
    for mtype in MODULES:
        assert rj.registry_has(mtype, "generic") and rj.registry_has(mtype, "mi355x"), mtype
    assert not rj.registry_has("fft", "no-such-provider")


@pytest.mark.parametrize("rows", [1, 5])
@pytest.mark.parametrize("forward", [True, False])
def test_fft_through_the_reference_registry_and_runtime(oracle, rows, forward):
    x = c1_tone(oracle, rows)
    axes = {"sample": 1, "batch": 0}
    cpu, hip = both("fft", {"forward": forward}, {"signal": (x, axes)}, "signal")
    assert_bit_equal(hip, cpu, "fft: provider mi355x vs the reference's pocketfft")
    assert np.abs(hip).max() > 100  # a spectrum, not zeros


def test_elementwise_modules_of_the_chain(oracle):
    rng = np.random.default_rng(7)
    x = c1_tone(oracle, 3)
    spec = both("fft", {}, {"signal": (x, {"sample": 1, "batch": 0})}, "signal")[0]
    cpu, hip = both("amplitude", {}, {"signal": (spec, {"sample": 1, "batch": 0})}, "signal")
    assert_bit_equal(hip, cpu, "amplitude")
    cpu_r, hip_r = both("range", {"min": -100.0, "max": 0.0}, {"signal": (cpu, {"sample": 1, "batch": 0})}, "signal")
    assert_bit_equal(hip_r, cpu_r, "range")
    w_cpu, w_hip = both("window", {"size": 4096}, {}, "window")
    assert_bit_equal(w_hip, w_cpu, "window")
    i_cpu, i_hip = both("invert", {}, {"signal": (w_cpu, {"sample": 0})}, "signal")
    assert_bit_equal(i_hip, i_cpu, "invert")
    a, b = cnoise(rng, 3, 4096), i_cpu.reshape(1, 4096)
    m_cpu, m_hip = both("multiply", {}, {"a": (a, {"sample": 1, "batch": 0}), "b": (b, {"sample": 1})}, "product")
    assert_bit_equal(m_hip, m_cpu, "multiply (broadcast window)")


def test_spectrogram_state_lives_on_the_device(oracle):
    rng = np.random.default_rng(9)
    states = []
    xs = [rng.uniform(-0.1, 1.1, (32, 512)).astype(np.float32) for _ in range(3)]
    for provider in ("generic", "mi355x"):
        with rj.RefModule("spectrogram", {"height": 64}, provider=provider) as m:
            m.input("signal", xs[0], sample=1, batch=0)
            assert m.start() == 0
            trace = []
            for x in xs:
                m.write("signal", x)
                assert m.compute() == 0
                trace.append(m.state("frequencyBins"))
            states.append(trace)
    for k, (cpu, hip) in enumerate(zip(*states)):
        assert_bit_equal(hip, cpu, f"spectrogram bins after cycle {k}")
    assert states[1][-1].max() > 0.03


def test_spectrum_engine_block_with_provider_mi355x(oracle):
    """The reference's own block expansion (spectrum_engine/block_impl.cc:120-217: cast -> window -> invert -> reshape ->
    multiply -> fft -> amplitude -> range) inside the reference's Flowgraph and scheduler, every module on the HIP library."""
    x = c1_tone(oracle, 5)
    outs = []
    for provider in ("generic", "mi355x"):
        with rj.RefFlowgraph() as fg:
            fg.source("src", x, sample=1, batch=0)
            assert fg.block("eng", "spectrum_engine", {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0},
                            {"buffer": "src:signal"}, provider=provider) == 0
            assert fg.state("eng") == 2, f"spectrum_engine ({provider}) did not reach CREATED"
            assert fg.compute() == 0 and fg.compute() == 0   # second cycle: the window chain has settled
            outs.append(np.array(fg.tensor("eng", "buffer")))
    assert_bit_equal(outs[1], outs[0], "spectrum_engine block: provider mi355x vs generic")
    ref = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
    assert_bit_equal(outs[1], ref, "... and vs the oracle")
