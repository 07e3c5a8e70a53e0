"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/jetstream_hip.h
declares, registers every hot-path module for (hip, native, generic), builds the reference's
twiddle table bit-exactly, mirrors the reference's validation errors, and FAILS LOUDLY (no CPU
fallback) when asked to compute without a GPU.  No kernel runs here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol(js):
    header = open(os.path.join(ROOT, "include", "jetstream_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(jst_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 45
    lib = C.CDLL(js.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_plugin_handshake_symbol(js):
    # include/jetstream/plugin.hh:48-87: {magic 0x4a535450, size 12, abi_version 1}
    lib = C.CDLL(js.LIB_PATH)
    abi = (C.c_uint32 * 3).in_dll(lib, "jetstream_plugin_abi")
    assert list(abi) == [0x4A535450, 12, 1]


def test_registry_has_the_hot_path_modules(js):
    listed = js.list_available_modules()
    for t in ("window", "invert", "reshape", "cast", "multiply", "multiply_constant", "fft",
              "amplitude", "range", "spectrogram", "waterfall", "ring_source"):
        assert f"{t}|hip|native|generic" in listed, t


def test_unknown_registration_is_an_error_not_a_fallback(js):
    with pytest.raises(js.JetstreamError, match="No module 'fft' registered"):
        js.Module("fft", {}, {}, provider="does-not-exist")
    with pytest.raises(js.JetstreamError, match="No module 'fft' registered"):
        js.Module("fft", {}, {}, device="cpu")  # the CPU path is the reference's, not ours


@pytest.mark.parametrize("m", range(0, 17))
def test_twiddle_table_bit_exact_vs_oracle(js, oracle, m):
    n = 1 << m
    mine, ref = js.fft_twiddles(n), oracle.fft_twiddles(n)
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def _host_view(js, array, **axes):
    """A tensor that CLAIMS to be on the HIP device but borrows host memory: good enough for
    validate(), which must not touch data or allocate (fft/module_impl.cc:15-22)."""
    t = js.Tensor.wrap(array.ctypes.data, array.nbytes, "hip",
                       {np.dtype(np.complex64): "CF32", np.dtype(np.float32): "F32"}[array.dtype],
                       array.shape)
    if axes:
        t.set_axes(**axes)
    return t


def test_validation_errors_mirror_the_reference(js):
    x = np.zeros((2, 14), np.complex64)
    with pytest.raises(js.JetstreamError, match=r"\[MODULE_FFT\] Input must contain valid signal axis"):
        js.Module("fft", {}, {"signal": _host_view(js, x)})  # rank 2 without axes (axis.cc:231-245)
    with pytest.raises(js.JetstreamError, match="Data type 'F64' is not implemented on the HIP device"):
        js.Module("fft", {}, {"signal": js.Tensor.wrap(x.ctypes.data, x.nbytes, "hip", "F64", (2, 14)).set_axes(sample=1, batch=0)})
    with pytest.raises(js.JetstreamError, match=r"\[MODULE_SPECTROGRAM\] Invalid height value"):
        js.Module("spectrogram", {"height": 0}, {})
    with pytest.raises(js.JetstreamError, match=r"\[MODULE_SPECTROGRAM\] Invalid height value"):
        js.Module("spectrogram", {"height": 4096}, {})
    with pytest.raises(js.JetstreamError, match=r"\[MODULE_WINDOW\] Window size cannot be zero"):
        js.Module("window", {"size": 0}, {})
    a = np.zeros((2, 3), np.complex64)
    b = np.zeros((4,), np.complex64)
    with pytest.raises(js.JetstreamError, match="are not broadcastable"):
        js.Module("multiply", {}, {"a": _host_view(js, a), "b": _host_view(js, b)})
    f = np.zeros((4, 5, 6), np.float32)
    with pytest.raises(js.JetstreamError, match="Unsupported auxiliary input axis"):
        js.Module("spectrogram", {}, {"signal": _host_view(js, f, sample=2, batch=0)})
    with pytest.raises(js.JetstreamError, match="cannot contain both sampleAxis and channelAxis"):
        js.Module("waterfall", {}, {"signal": _host_view(js, f[0], sample=1, channel=0)})
    with pytest.raises(js.JetstreamError, match="sampleAxis or channelAxis metadata"):
        js.Module("amplitude", {}, {"signal": _host_view(js, f[0])})
    with pytest.raises(js.JetstreamError, match=r"requested missing input 'signal'"):
        js.Module("fft", {}, {})
    with pytest.raises(js.JetstreamError, match="Shape must use bracket notation"):
        js.Module("reshape", {"shape": "4,4"}, {"buffer": _host_view(js, a)})


def test_views_strides_and_axes_are_in_elements(js):
    base = np.arange(4 * 6 * 8, dtype=np.float32)
    t = js.Tensor.wrap(base.ctypes.data, base.nbytes, "cpu", "F32", (4, 6, 8))
    assert t.stride == (48, 8, 1) and t.offset == 0
    t.slice(1, 2, 6, 2)
    assert t.shape == (4, 2, 8) and t.stride == (48, 16, 1) and t.offset == 16
    t.permute((2, 0, 1))
    assert t.shape == (8, 4, 2) and t.stride == (1, 48, 16)
    with pytest.raises(js.JetstreamError, match="non-contiguous"):
        t.reshape((64,))
    u = js.Tensor.wrap(base.ctypes.data, base.nbytes, "cpu", "F32", (8,))
    u.expand_dims(0).broadcast_to((3, 8))
    assert u.shape == (3, 8) and u.stride == (0, 1)
    u.set_axes(sample=1, batch=0)
    assert u.axes == {"sample": 1, "batch": 0, "channel": None}
    with pytest.raises(js.JetstreamError, match="exceeds"):
        js.Tensor.wrap(base.ctypes.data, 16, "cpu", "F32", (8,))


def test_no_gpu_means_loud_failure(js):
    if js.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(js.JetstreamError, match="hipMalloc"):
        js.Tensor.create("hip", "CF32", (8, 64))
    x = np.zeros((1, 64), np.complex64)
    with pytest.raises(js.JetstreamError):  # create() must allocate its output on the device
        js.Module("fft", {}, {"signal": _host_view(js, x, sample=1, batch=0)})


def test_runtime_flags_agree_between_header_and_mirror(js):
    """The Python mirror's RUNTIME_* constants are the header's JST_RUNTIME_* enumerators (cycle batching included)."""
    header = open(os.path.join(ROOT, "include", "jetstream_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    flags = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r"JST_RUNTIME_([A-Z]+)\s*=\s*1\s*<<\s*(\d+)", header)}
    assert set(flags) == {"GRAPH", "FUSE", "TIMING", "PIPELINE", "COMBINE", "BATCH"}, flags
    for name, value in flags.items():
        assert getattr(js, "RUNTIME_" + name) == value, name


def test_batching_needs_a_gpu_but_the_flag_is_understood(js):
    """No GPU here: asking for a batched runtime must fail loudly like every other compute request (no CPU fallback),
    and the two symbols the bench reads the launch form from must be there."""
    lib = C.CDLL(js.LIB_PATH)
    assert hasattr(lib, "jst_runtime_batched") and hasattr(lib, "jst_runtime_unit_mean_cycles")
    lib.jst_runtime_batched.restype = C.c_int
    lib.jst_runtime_unit_mean_cycles.restype = C.c_double
    assert lib.jst_runtime_batched(None) == 0
    assert lib.jst_runtime_unit_mean_cycles(None, b"spectrum_fused") < 0


def test_comm_symbols_and_single_rank_semantics(js):
    """The collective behind the C ABI (csrc/jst/comm.cc): a one-rank communicator needs no RCCL and no GPU; more ranks
    without rank 0's id fail loudly with the reason in jst_last_error (operand checks: tests/test_gpu_advice_r04.py)."""
    assert js.comm_available() in (True, False)
    c = js.Comm(0, 1, None)
    assert (c.rank, c.world, c.calls, c.uses_rccl) == (0, 1, 0, False)
    with pytest.raises(js.JetstreamError, match="unique id"):
        js.Comm(0, 2, None)
    with pytest.raises(js.JetstreamError, match="invalid rank"):
        js.Comm(3, 2, bytes(128))
