"""GPU: the REFERENCE runs DEVICE-RESIDENT on the library.  oracle/_ref/libref_jetstream_devhip.so is the reference's own core
compiled with integration/device_hip/core_hip_device.patch (DeviceType::HIP in the enum and name maps, MakeBackend, the Runtime
factory, the CPU mirror, duplicate) plus the reference-side units of integration/device_hip/: the HBM buffer backend, the HIP
segment runtime and one module_impl_native_hip per module of the spectrum chain -- the reference's own Impl (validate / define
/ create: output tensors allocated ON THE DEVICE through Tensor::create(device(), ..)) with the compute hooks on a library
module that works in place on those tensors.  Here the reference's
    Flowgraph::blockCreate(.., DeviceType::HIP) -> Scheduler (src/scheduler_synchronous.cc:534-749: order, static settlement,
    runtime segments per device) -> Runtime(HIP)::compute (src/runtime/runtime.cc:17-61 through the patch)
drive the HIP kernels with NO host copy between modules, and -- a segment of library modules being handed to ONE jst_runtime
(integration/device_hip/runtime_native_hip_impl.cc) -- through hipGraph capture and the fused kernels.  Outputs and Spectrogram
bins are compared BIT FOR BIT with the same reference calls on DeviceType::CPU (its own CPU modules)."""
import numpy as np
import pytest

from oracle import ref_jetstream as rj
from util import assert_bit_equal

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not rj.device_hip_library_available(), reason="oracle/_ref/libref_jetstream_devhip.so not built")]

CHAIN = ("cast", "window", "invert", "reshape", "multiply", "fft", "amplitude", "range", "spectrogram", "ring_source")
ENGINE = {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0}


@pytest.fixture(scope="module", autouse=True)
def reference_with_the_hip_device(js):  # `js`: the product library is loaded (and the device selected) first
    rj.use_device_hip_library()
    rj.hip_runtime_configure(True, 0)
    yield
    rj.hip_runtime_configure(True, 0)


def tones(oracle, rows, n=4096, seed=1234, sigma=1e-3, scale=1.0):
    rng = np.random.default_rng(seed)
    fs = 2.0e6
    x = np.empty((rows, n), np.complex64)
    for r in range(rows):
        x[r], _ = oracle.signal_cosine(n, 1.0, (100.25 + r) * fs / n, fs)
    x += ((rng.standard_normal((rows, n)) + 1j * rng.standard_normal((rows, n))) * sigma).astype(np.complex64)
    return (x * np.float32(scale)).astype(np.complex64)


def cpu_flowgraph(cycles_in, height, provider="generic"):
    """The reference on DeviceType::CPU: source -> spectrum_engine -> spectrogram, one compute per entry of cycles_in; returns
    the engine's output after the last cycle and the Spectrogram's bins."""
    with rj.RefFlowgraph() as fg:
        src = fg.source("src", cycles_in[0], sample=1, batch=0)
        assert fg.block("eng", "spectrum_engine", ENGINE, {"buffer": "src:signal"}) == 0
        assert fg.block("spec", "spectrogram", {"height": height}, {"signal": "eng:buffer"}) == 0
        for x in cycles_in:
            src[...] = x
            assert fg.compute() == 0
        return np.array(fg.tensor("eng", "buffer")), cpu_bins(cycles_in, height)


def cpu_bins(cycles_in, height):
    """Spectrogram state through the reference's module API (a flowgraph does not expose module state)."""
    from oracle import oracle
    bins = None
    with rj.RefModule("spectrogram", {"height": height}) as m:
        first = oracle.spectrum_chain(cycles_in[0], -100.0, 0.0)["range"]
        m.input("signal", first, sample=1, batch=0)
        assert m.start() == 0
        for x in cycles_in:
            m.write("signal", oracle.spectrum_chain(x, -100.0, 0.0)["range"])
            assert m.compute() == 0
        bins = m.state("frequencyBins")
    return bins


def test_the_registry_holds_the_device():
    for mtype in CHAIN:
        assert rj.registry_has(mtype, "generic", device="hip"), mtype
    for mtype in CHAIN[:-1]:
        assert rj.registry_has(mtype, "generic", device="cpu") or mtype == "ring_source"
    assert not rj.registry_has("fft", "generic", device="cuda")      # this build has no CUDA modules


@pytest.mark.parametrize("forward", [True, False])
def test_fft_module_on_the_device(oracle, forward):
    """Registry::BuildModule(fft, HIP, NATIVE) -> Module::create (the input a HIP tensor, the output allocated by the
    reference through buffer_hip.cc) -> Runtime(HIP)::compute, against the same calls on the CPU."""
    x = tones(oracle, 5)
    outs = []
    for device in ("cpu", "hip"):
        with rj.RefModule("fft", {"forward": forward}, device=device) as m:
            m.input("signal", x, sample=1, batch=0)
            assert m.start() == 0, f"fft ({device}): Module::create failed"
            assert m.compute() == 0
            if device == "hip":
                d = rj._Desc()
                assert m._l.ref_mod_output(m._h, b"signal", rj.C.byref(d)) == 0
                assert int(d.device) == rj.DEVICE_HIP and m._l.ref_dev_pointer_kind(rj.C.c_void_p(d.data)) == 2   # HBM
            outs.append(m.output("signal"))
    assert_bit_equal(outs[1], outs[0], "fft on DeviceType::HIP vs the reference's pocketfft")
    assert np.abs(outs[1]).max() > 100


@pytest.mark.parametrize("hand_off", [True, False, 2, 3])
def test_spectrum_engine_and_spectrogram_device_resident(oracle, hand_off):
    """ring_source -> spectrum_engine -> spectrogram, every block on DeviceType::HIP, inside the reference's Flowgraph and
    scheduler.  hand_off: the HIP runtime gives the segment to one jst_runtime (fusion; direct launches for synchronous cycles --
    the default --, 2: direct launches always, 3: replayed from a hipGraph) -- or submits module by module on its own stream
    (the CUDA runtime's shape).  Either way: bit-equal to the CPU device."""
    rows, n, h = 16, 4096, 256
    xs = [tones(oracle, rows, seed=40 + c, scale=0.5 + 0.25 * c) for c in range(3)]
    want_out, want_bins = cpu_flowgraph(xs, h)
    rj.hip_runtime_configure(hand_off, 0)
    with rj.RefFlowgraph() as fg:
        assert fg.ring_source("src", rows, n, 1) == 0
        assert fg.block("eng", "spectrum_engine", ENGINE, {"buffer": "src:buffer"}, device="hip") == 0
        assert fg.state("eng") == 2, "spectrum_engine (hip) did not reach CREATED"
        assert fg.block("spec", "spectrogram", {"height": h}, {"signal": "eng:buffer"}, device="hip") == 0
        for x in xs:
            fg.ring_write("src", 0, x)
            assert fg.compute() == 0
        units = rj.hip_runtime_units()
        d = fg.desc("eng", "buffer")
        assert int(d.device) == rj.DEVICE_HIP and fg._l.ref_dev_pointer_kind(rj.C.c_void_p(d.data)) == 2   # the output lives in HBM
        got_out = np.array(fg.tensor("eng", "buffer"))
        got_bins = np.array(rj.hip_directory("spec-spectrogram", "state:frequencyBins"))
    rj.hip_runtime_configure(True, 0)
    if hand_off:
        assert "spectrum_fused(" in units and "+indices" in units, units     # ONE kernel for multiply + fft + amplitude + range
    else:
        assert units == "", units
    assert_bit_equal(got_out, want_out, "spectrum_engine on DeviceType::HIP vs DeviceType::CPU")
    assert_bit_equal(got_bins.reshape(-1), want_bins.reshape(-1), "Spectrogram bins on DeviceType::HIP vs DeviceType::CPU")
    assert got_bins.max() > 0.03


def test_cpu_source_crosses_through_duplicate(oracle):
    """An existing flowgraph with `device: hip` on the nodes of the path: the CPU source's tensor crosses once, through the
    reference's own `duplicate{outputDevice=hip}` (a CPU module writing the CPU mirror of a host-accessible HIP buffer,
    src/domains/core/duplicate/module_impl.cc:80-110 + the patch's buffer_cpu.cc branch)."""
    rows, h = 8, 64
    xs = [tones(oracle, rows, seed=60 + c) for c in range(2)]
    want_out, want_bins = cpu_flowgraph(xs, h)
    with rj.RefFlowgraph() as fg:
        src = fg.source("src", xs[0], sample=1, batch=0)
        assert fg.block("dup", "duplicate", {"outputDevice": "hip", "hostAccessible": True}, {"buffer": "src:signal"}) == 0
        assert fg.block("eng", "spectrum_engine", ENGINE, {"buffer": "dup:buffer"}, device="hip") == 0
        assert fg.block("spec", "spectrogram", {"height": h}, {"signal": "eng:buffer"}, device="hip") == 0
        for x in xs:
            src[...] = x
            assert fg.compute() == 0
        assert int(fg.desc("dup", "buffer").device) == rj.DEVICE_HIP
        got_out = np.array(fg.tensor("eng", "buffer"))
        got_bins = np.array(rj.hip_directory("spec-spectrogram", "state:frequencyBins"))
    assert_bit_equal(got_out, want_out, "CPU source -> duplicate -> HIP chain")
    assert_bit_equal(got_bins.reshape(-1), want_bins.reshape(-1), "... and its Spectrogram bins")


def test_deferred_cycles_run_as_spans(oracle):
    """Deferred cycles (opt-in): with a RESIDENT ring the HIP runtime counts the scheduler's cycles and runs them as one
    cycle-batched span per flush -- one launch per unit for `defer` cycles.  What a reader sees after the flush is what the
    synchronous cycles leave: the last cycle's output, the bins after every cycle's decay and hits."""
    rows, n, h, slots, cycles = 16, 4096, 256, 4, 11
    ring = [tones(oracle, rows, seed=80 + s, scale=0.4 + 0.2 * s) for s in range(slots)]
    xs = [ring[c % slots] for c in range(cycles)]
    want_out, want_bins = cpu_flowgraph(xs, h)
    rj.hip_runtime_configure(True, slots)
    try:
        with rj.RefFlowgraph() as fg:
            assert fg.ring_source("src", rows, n, slots) == 0
            for s in range(slots):
                fg.ring_write("src", s, ring[s])
            assert fg.block("eng", "spectrum_engine", ENGINE, {"buffer": "src:buffer"}, device="hip") == 0
            assert fg.block("spec", "spectrogram", {"height": h}, {"signal": "eng:buffer"}, device="hip") == 0
            for _ in range(cycles):
                assert fg.compute() == 0
            rj.hip_runtime_flush()
            units = rj.hip_runtime_units()
            got_out = np.array(fg.tensor("eng", "buffer"))
            got_bins = np.array(rj.hip_directory("spec-spectrogram", "state:frequencyBins"))
    finally:
        rj.hip_runtime_configure(True, 0)
    assert "[batched]" in units and "+indices" in units, units
    assert_bit_equal(got_out, want_out, "deferred spans: the engine's output after the last cycle")
    assert_bit_equal(got_bins.reshape(-1), want_bins.reshape(-1), "deferred spans: Spectrogram bins")


# ---- the side chains: Filter block, slice, FM, the AGC'd spectrum engine of multi-fm.yml, the sinks with state ---------------------
def multi_fm_flowgraph(fg, device, source_block, source_port):
    """The chains of the reference's examples/flowgraphs/multi-fm.yml behind a source: filter (2 heads at +-400 kHz, /10) ->
    slice [:, 1, :] -> spectrum_engine{enableAgc}; slice [:, 0, :] -> fm -> decimator; every block on `device`."""
    kw = {} if device == "cpu" else {"device": device}
    src = f"{source_block}:{source_port}"
    assert fg.block("flt", "filter", {"taps": 51, "heads": 2, "center": [400000.0, -400000.0], "bandwidth": 200000.0, "sampleRate": 2000000.0},
                    {"signal": src}, **kw) == 0
    assert fg.state("flt") == 2, f"filter ({device}) did not reach CREATED"
    assert fg.block("slice", "slice", {"contiguous": True, "slice": "[:, 1, :]"}, {"buffer": "flt:buffer"}, **kw) == 0
    assert fg.block("spectrum_engine", "spectrum_engine", {"rangeMin": -270.0, "enableAgc": True, "rangeMax": 1.0, "enableScale": True},
                    {"buffer": "slice:buffer"}, **kw) == 0
    assert fg.block("sli23", "slice", {"contiguous": True, "slice": "[:, 0, :]"}, {"buffer": "flt:buffer"}, **kw) == 0
    assert fg.block("fm", "fm", {"sampleRate": 200000.0}, {"signal": "sli23:buffer"}, **kw) == 0
    assert fg.block("dec", "decimator", {"ratio": 4}, {"buffer": "fm:signal"}, **kw) == 0
    for name in ("slice", "spectrum_engine", "sli23", "fm", "dec"):
        assert fg.state(name) == 2, f"{name} ({device}) did not reach CREATED"


def test_filter_block_and_fm_chain_device_resident(oracle):
    """The reference's own `filter` block (src/domains/dsp/filter/block_impl.cc:350-582: filter_taps, cast, expand_dims, pad x 2,
    fft x 2, reshape, multiply, fold, ifft, multiply_constant, phase_correction, unpad, overlap_add), `slice` (slice +
    duplicate), `fm`, `decimator` (reshape + arithmetic) and the AGC'd `spectrum_engine` of multi-fm.yml, every module on
    DeviceType::HIP inside the reference's scheduler, handed to ONE library runtime (fused Filter head and tail) -- against
    the same blocks on DeviceType::CPU, three cycles (overlap, phase and demodulator state carried), bit for bit."""
    rng = np.random.default_rng(91)
    b, n = 8, 8000
    xs = [((rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))) * 0.05).astype(np.complex64) for _ in range(3)]
    ports = [("flt", "buffer"), ("fm", "signal"), ("dec", "buffer"), ("spectrum_engine", "buffer")]
    want, got = [], []
    with rj.RefFlowgraph() as fg:
        src = fg.source("src", xs[0], sample=1, batch=0)
        fg.set_attr("src", "signal", "sampleRate", rj.ATTR_F32, 2.0e6)
        fg.set_attr("src", "signal", "frequency", rj.ATTR_F32, 96.9e6)
        multi_fm_flowgraph(fg, "cpu", "src", "signal")
        for x in xs:
            src[...] = x
            assert fg.compute() == 0
            want.append([np.array(fg.tensor(*p)) for p in ports])
    with rj.RefFlowgraph() as fg:
        assert fg.ring_source("src", b, n, 1) == 0
        multi_fm_flowgraph(fg, "hip", "src", "buffer")
        for x in xs:
            fg.ring_write("src", 0, x)
            assert fg.compute() == 0
            got.append([np.array(fg.tensor(*p)) for p in ports])
        units = rj.hip_runtime_units()
    # the Filter's fused head and tail, the elided copies behind the slices, the AGC chain: fused from the reference's side
    for unit in ("fft_padded_fold(", "ifft_phase_unpad_overlap(", "duplicate(elided)", "fft_windowed(", "agc_amplitude_range("):
        assert unit in units, (unit, units)
    for c, (w, g) in enumerate(zip(want, got)):
        for (block, port), a, d in zip(ports, w, g):
            assert_bit_equal(d, a, f"{block}:{port} on DeviceType::HIP vs DeviceType::CPU, cycle {c}")
    assert np.abs(got[-1][0]).max() > 1e-3 and np.isfinite(got[-1][1]).all()


@pytest.mark.parametrize("mtype,cfg,state,shape", [("waterfall", {"height": 48}, "frequencyBins", (20, 512)),
                                                   ("waterfall", {"height": 16}, "frequencyBins", (40, 256)),   # more batches than rows
                                                   ("lineplot", {"averaging": 4, "decimation": 2}, "signalPoints", (16, 1024))])
def test_sinks_with_state_on_the_device(mtype, cfg, state, shape):
    """waterfall / lineplot as (DeviceType::HIP, NATIVE) modules: the library's state lives in the reference module's own state
    tensor (what its present half reads), bit-equal to the CPU module's over three submissions."""
    rng = np.random.default_rng(17)
    xs = [rng.uniform(0.0, 1.0, shape).astype(np.float32) for _ in range(3)]
    states = []
    for device in ("cpu", "hip"):
        with rj.RefModule(mtype, cfg, device=device) as m:
            m.input("signal", xs[0], sample=1, batch=0)
            assert m.start() == 0, f"{mtype} ({device}): Module::create failed"
            trace = []
            for x in xs:
                m.write("signal", x)
                assert m.compute() == 0
                trace.append(np.array(m.state(state)))
            states.append(trace)
    for k, (cpu, hip) in enumerate(zip(*states)):
        assert_bit_equal(hip, cpu, f"{mtype} {state} after submission {k}")
