"""numpy front-end for the CPU oracle (oracle/jst_oracle.c) and the compiled reference
pocketfft (oracle/_ref/libref_pocketfft.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (cyberether_amd) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjst_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_pocketfft.so")
_REF_HELPERS_PATH = os.path.join(_HERE, "_ref", "libref_helpers.so")
_REF_JETSTREAM_PATH = os.path.join(_HERE, "_ref", "libref_jetstream.so")
# the same reference objects linked with integration/mi355x_provider/ against cyberether_amd/lib/libjetstream_hip.so
_REF_JETSTREAM_HIP_PATH = os.path.join(_HERE, "_ref", "libref_jetstream_hip.so")

_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)


def build(force: bool = False) -> None:
    """Compile the restatement (and, when /root/reference exists, the reference pocketfft)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "jst_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libjst_oracle.so"], stdout=subprocess.DEVNULL)
    if (force or not os.path.exists(_REF_PATH) or not os.path.exists(_REF_HELPERS_PATH)
            or not os.path.exists(_REF_JETSTREAM_PATH) or not os.path.exists(_REF_JETSTREAM_HIP_PATH)) and os.path.exists(
        "/root/reference/src/domains/dsp/fft/pocketfft.hh"
    ):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.jst_oracle_approx_log10.restype = C.c_float
        _lib.jst_oracle_approx_log10.argtypes = [C.c_float]
        _lib.jst_oracle_amplitude_coeff.restype = C.c_float
        _lib.jst_oracle_amplitude_coeff.argtypes = [C.c_uint64]
        _lib.jst_oracle_spectrogram_decay.restype = C.c_float
        _lib.jst_oracle_spectrogram_decay.argtypes = [C.c_uint64]
        _lib.jst_oracle_cast_u64.restype = C.c_uint64
        _lib.jst_oracle_cast_u64.argtypes = [C.c_float]
        _lib.jst_oracle_fft_c2c.restype = C.c_int
        _lib.jst_oracle_fm_state_size.restype = C.c_uint64
        _lib.jst_oracle_fm_coeffs_size.restype = C.c_uint64
    return _lib


def have_ref() -> bool:
    build()
    return os.path.exists(_REF_PATH)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(_REF_PATH)
        for name in ("ref_fft_c2c", "ref_fft_r2c", "ref_fft_r2r_fftpack"):
            getattr(_ref, name).restype = C.c_int
    return _ref


_ref_helpers = None


def have_ref_helpers() -> bool:
    build()
    return os.path.exists(_REF_HELPERS_PATH)


def ref_helpers() -> C.CDLL:
    """The reference's own inline ApproxLog10 and waterfall ring arithmetic (oracle/ref_helpers.cc, compiled in place)."""
    global _ref_helpers
    if _ref_helpers is None:
        build()
        _ref_helpers = C.CDLL(_REF_HELPERS_PATH)
        _ref_helpers.ref_approx_log10.restype = C.c_float
        _ref_helpers.ref_approx_log10.argtypes = [C.c_float]
    return _ref_helpers


def _p(a: np.ndarray, t=_f32p):
    return a.ctypes.data_as(t)


def _u64(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.uint64))


def _elem_strides(a: np.ndarray) -> np.ndarray:
    return _u64([s // a.itemsize for s in a.strides])


# ----------------------------------------------------------------------------------- window
def window(n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.complex64)
    lib().jst_oracle_window(_p(out), C.c_uint64(n))
    return out


# ----------------------------------------------------------------------------------- invert
def invert(x: np.ndarray, axis: int = -1) -> np.ndarray:
    """x: contiguous F32 or CF32 array; axis = resolved sample axis."""
    x = np.ascontiguousarray(x)
    axis = axis % x.ndim
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.uint64)) if axis + 1 < x.ndim else 1
    length = x.shape[axis]
    out = np.empty(x.shape, dtype=np.complex64)
    if x.dtype == np.complex64:
        lib().jst_oracle_invert_cf32(_p(x), _p(out), C.c_uint64(x.size), C.c_uint64(inner),
                                     C.c_uint64(length))
    elif x.dtype == np.float32:
        lib().jst_oracle_invert_f32(_p(x), _p(out), C.c_uint64(x.size), C.c_uint64(inner),
                                    C.c_uint64(length))
    else:
        raise TypeError(x.dtype)
    return out


# --------------------------------------------------------------------------------- multiply
def multiply(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """numpy-style right-aligned broadcast product, reference arithmetic (no FMA, __mulsc3)."""
    shape = np.broadcast_shapes(a.shape, b.shape)
    av, bv = np.broadcast_to(a, shape), np.broadcast_to(b, shape)
    out = np.empty(shape, dtype=a.dtype)
    sh = _u64(shape)
    fn = {np.dtype(np.complex64): lib().jst_oracle_multiply_cf32,
          np.dtype(np.float32): lib().jst_oracle_multiply_f32}[a.dtype]
    sa, sb, sc = _elem_strides(av), _elem_strides(bv), _elem_strides(out)
    fn(C.c_uint32(len(shape)), _p(sh, _u64p), _p(av), _p(sa, _u64p), _p(bv), _p(sb, _u64p),
       _p(out), _p(sc, _u64p))
    return out


# -------------------------------------------------------------------------------------- fft
def fft_factors(n: int):
    f = (C.c_uint32 * 64)()
    nf = lib().jst_oracle_fft_factors(C.c_uint64(n), f)
    return None if nf < 0 else [int(f[i]) for i in range(nf)]


def fft_twiddles(n: int) -> np.ndarray:
    tw = np.empty(n, dtype=np.complex64)
    lib().jst_oracle_fft_twiddles(_p(tw), C.c_uint64(n))
    return tw


def fft_c2c(x: np.ndarray, forward: bool = True) -> np.ndarray:
    """Restated pocketfft c2c over the LAST axis: any length (radix 2/3/4/5/7/8/11 passes, the
    generic odd-radix pass, and Bluestein when pocketfft_c would pick it)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n = x.shape[-1]
    batch = x.size // n if n else 0
    out = np.empty_like(x)
    rc = lib().jst_oracle_fft_c2c(_p(x), _p(out), C.c_uint64(n), C.c_uint64(batch),
                                  C.c_int(1 if forward else 0))
    if rc != 0:
        raise ValueError(f"zero-length FFT requested (n={n})")
    return out


def fft_r2r(x: np.ndarray, forward: bool = True) -> np.ndarray:
    """Restated pocketfft r2r_fftpack over the LAST axis (halfcomplex FFTPACK format): radices
    2/3/4/5, the generic radix (radfg/radbg) for larger prime factors, and the real Bluestein path."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[-1]
    out = np.empty_like(x)
    rc = lib().jst_oracle_fft_r2r(_p(x), _p(out), C.c_uint64(n), C.c_uint64(x.size // n if n else 0),
                                  C.c_int(1 if forward else 0))
    if rc != 0:
        raise ValueError("zero-length FFT requested")
    return out


def fft_r2c(x: np.ndarray) -> np.ndarray:
    """Restated pocketfft r2c (forward) over the LAST axis: n reals -> n//2+1 complex."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[-1]
    out = np.empty(x.shape[:-1] + (n // 2 + 1,), np.complex64)
    rc = lib().jst_oracle_fft_r2c(_p(x), _p(out), C.c_uint64(n), C.c_uint64(x.size // n if n else 0))
    if rc != 0:
        raise ValueError("zero-length FFT requested")
    return out


def rfft_bluestein_size(n: int) -> int:
    lib().jst_oracle_rfft_bluestein_size.restype = C.c_uint64
    return int(lib().jst_oracle_rfft_bluestein_size(C.c_uint64(n)))


def fft_bluestein_size(n: int) -> int:
    """0 when pocketfft_c plans a cfftp of n, else the Bluestein convolution length n2."""
    lib().jst_oracle_fft_bluestein_size.restype = C.c_uint64
    return int(lib().jst_oracle_fft_bluestein_size(C.c_uint64(n)))


def _ref_call(fn, x: np.ndarray, out: np.ndarray, axis: int, forward: bool) -> np.ndarray:
    sh = _u64(x.shape)
    si = np.ascontiguousarray(np.asarray(x.strides, dtype=np.int64))
    so = np.ascontiguousarray(np.asarray(out.strides, dtype=np.int64))
    rc = fn(C.c_uint32(x.ndim), _p(sh, _u64p), _p(si, _i64p), _p(so, _i64p),
            C.c_uint64(axis % x.ndim), C.c_int(1 if forward else 0), _p(x), _p(out))
    if rc != 0:
        raise RuntimeError("reference pocketfft raised")
    return out


def ref_fft_c2c(x: np.ndarray, axis: int = -1, forward: bool = True) -> np.ndarray:
    """The reference's own pocketfft (compiled from /root/reference), any length/strides."""
    assert x.dtype == np.complex64
    return _ref_call(ref().ref_fft_c2c, x, np.empty(x.shape, np.complex64), axis, forward)


def ref_fft_r2c(x: np.ndarray, axis: int = -1) -> np.ndarray:
    assert x.dtype == np.float32
    axis = axis % x.ndim
    oshape = list(x.shape)
    oshape[axis] = x.shape[axis] // 2 + 1
    return _ref_call(ref().ref_fft_r2c, x, np.empty(oshape, np.complex64), axis, True)


def ref_fft_r2r(x: np.ndarray, axis: int = -1, forward: bool = True) -> np.ndarray:
    assert x.dtype == np.float32
    return _ref_call(ref().ref_fft_r2r_fftpack, x, np.empty(x.shape, np.float32), axis, forward)


# -------------------------------------------------------------------------------- amplitude
def approx_log10(x: float) -> float:
    return float(lib().jst_oracle_approx_log10(C.c_float(x)))


def amplitude_coeff(n: int) -> float:
    return float(lib().jst_oracle_amplitude_coeff(C.c_uint64(n)))


def amplitude(x: np.ndarray, norm_size: int) -> np.ndarray:
    """x: F32 or CF32 (any strides); norm_size = extent of the sample axis (1 if channel-only)."""
    out = np.empty(x.shape, dtype=np.float32)
    coeff = C.c_float(amplitude_coeff(norm_size))
    sh, si, so = _u64(x.shape), _elem_strides(x), _elem_strides(out)
    fn = {np.dtype(np.complex64): lib().jst_oracle_amplitude_cf32,
          np.dtype(np.float32): lib().jst_oracle_amplitude_f32}[x.dtype]
    fn(C.c_uint32(x.ndim), _p(sh, _u64p), _p(x), _p(si, _u64p), _p(out), _p(so, _u64p), coeff)
    return out


# ------------------------------------------------------------------------------------ range
def range_coeffs(vmin: float, vmax: float):
    s, o = C.c_float(), C.c_float()
    lib().jst_oracle_range_coeffs(C.c_float(vmin), C.c_float(vmax), C.byref(s), C.byref(o))
    return float(s.value), float(o.value)


def range_(x: np.ndarray, vmin: float, vmax: float) -> np.ndarray:
    assert x.dtype == np.float32
    s, o = range_coeffs(vmin, vmax)
    out = np.empty(x.shape, dtype=np.float32)
    sh, si, so = _u64(x.shape), _elem_strides(x), _elem_strides(out)
    lib().jst_oracle_range_f32(C.c_uint32(x.ndim), _p(sh, _u64p), _p(x), _p(si, _u64p), _p(out),
                               _p(so, _u64p), C.c_float(s), C.c_float(o))
    return out


# ------------------------------------------------------------------------------ spectrogram
def spectrogram_decay(batches: int) -> float:
    return float(lib().jst_oracle_spectrogram_decay(C.c_uint64(batches)))


def spectrogram(bins: np.ndarray, x: np.ndarray, height: int, batch_axis=0, elem_axis=1) -> None:
    """In-place update of bins (F32, flat [height*width], layout [height][width]).
    x: F32 rank-1 (no batch axis: batch_axis=None) or rank-2 with the given axes."""
    assert x.dtype == np.float32 and bins.dtype == np.float32 and bins.flags.c_contiguous
    if x.ndim == 1 or batch_axis is None:
        batches, bstride = 1, 0
        width, estride = x.shape[-1], x.strides[-1] // 4
    else:
        batches, bstride = x.shape[batch_axis], x.strides[batch_axis] // 4
        width, estride = x.shape[elem_axis], x.strides[elem_axis] // 4
    assert bins.size == width * height
    lib().jst_oracle_spectrogram(_p(bins), _p(x), C.c_uint64(batches), C.c_uint64(width),
                                 C.c_uint64(height), C.c_uint64(bstride), C.c_uint64(estride),
                                 C.c_float(spectrogram_decay(batches)))


# -------------------------------------------------------------------------------- waterfall
def waterfall_plan(write_index: int, batches: int, height: int):
    p = (C.c_uint64 * 3)()
    lib().jst_oracle_waterfall_plan(C.c_uint64(write_index), C.c_uint64(batches),
                                    C.c_uint64(height), p)
    return tuple(int(v) for v in p)


def waterfall_advance(state, batches: int, height: int):
    s = (C.c_uint64 * 2)(*state)
    lib().jst_oracle_waterfall_advance(s, C.c_uint64(batches), C.c_uint64(height))
    return (int(s[0]), int(s[1]))


def waterfall_dirty_plan(state, height: int):
    s = (C.c_uint64 * 2)(*state)
    p = (C.c_uint64 * 3)()
    lib().jst_oracle_waterfall_dirty_plan(s, C.c_uint64(height), p)
    return tuple(int(v) for v in p)


def waterfall(bins: np.ndarray, state, x: np.ndarray, height: int):
    """bins F32 [height, width] updated in place; returns the new (writeIndex, dirtyRows)."""
    assert x.dtype == np.float32 and x.ndim == 2 and bins.flags.c_contiguous
    s = (C.c_uint64 * 2)(*state)
    lib().jst_oracle_waterfall(_p(bins), s, _p(x), C.c_uint64(x.shape[0]), C.c_uint64(x.shape[1]),
                               C.c_uint64(height), C.c_uint64(x.strides[0] // 4),
                               C.c_uint64(x.strides[1] // 4))
    return (int(s[0]), int(s[1]))


# ------------------------------------------------------------------------- signal generator
def signal_cosine(count: int, amplitude_: float, frequency: float, sample_rate: float,
                  dc_offset: float = 0.0, phase: float = 0.0):
    out = np.empty(count, dtype=np.complex64)
    ph = C.c_double(phase)
    lib().jst_oracle_signal_cosine_cf32(_p(out), C.c_uint64(count), C.c_double(amplitude_),
                                        C.c_double(frequency), C.c_double(sample_rate),
                                        C.c_double(dc_offset), C.byref(ph))
    return out, float(ph.value)


# ------------------------------------------------------------------------- composed chains
_SHAPES = {"sine": 0, "cosine": 1, "square": 2, "triangle": 3, "sawtooth": 4, "dc": 5, "chirp": 6}


def signal(shape: str, count: int, complex_out: bool, state, amplitude_=1.0, frequency=1000.0,
           sample_rate=1.0e6, dc_offset=0.0, chirp_start=1000.0, chirp_end=10000.0, chirp_duration=1.0):
    """Any deterministic waveform of the signal generator; state = [phase, chirpTime] is carried
    (returned updated).  Returns (samples, new_state)."""
    out = np.empty(count, dtype=np.complex64 if complex_out else np.float32)
    st = (C.c_double * 2)(float(state[0]), float(state[1]))
    lib().jst_oracle_signal(_p(out.view(np.float32)), C.c_uint64(count), C.c_int(1 if complex_out else 0),
                            C.c_int(_SHAPES[shape]), C.c_double(amplitude_), C.c_double(frequency),
                            C.c_double(sample_rate), C.c_double(dc_offset), C.c_double(chirp_start),
                            C.c_double(chirp_end), C.c_double(chirp_duration), st)
    return out, [st[0], st[1]]


def spectrum_chain(x: np.ndarray, range_min=None, range_max=None):
    """Window -> Invert -> (reshape) -> Multiply -> FFT -> Amplitude -> [Range] over the last axis
    (src/domains/dsp/spectrum_engine/block_impl.cc:120-217).  Returns dict of every stage."""
    n = x.shape[-1]
    w = invert(window(n))
    prod = multiply(x, w.reshape((1,) * (x.ndim - 1) + (n,)))
    spec = fft_c2c(prod, True)
    amp = amplitude(spec, n)
    out = {"window": w, "product": prod, "fft": spec, "amplitude": amp}
    if range_min is not None:
        out["range"] = range_(amp, range_min, range_max)
    return out



def chain_pass(x: np.ndarray, bins: np.ndarray, height: int, range_min: float = -100.0, range_max: float = 0.0,
               use_ref: bool = False) -> np.ndarray:
    """One compute cycle of the headline chain on a dense CF32[rows, n] batch in dense C loops
    (jst_oracle.c: jst_oracle_chain_pass = multiply by the inverted window -> FFT -> amplitude -> range ->
    spectrogram update of `bins` in place): the same arithmetic as spectrum_chain() + spectrogram(), ~50 MS/s,
    so the FULL configs[1] batch (1024 x 4096) is checked in 80 ms per cycle.  Returns the range output."""
    assert x.dtype == np.complex64 and x.ndim == 2 and x.flags.c_contiguous
    assert bins.dtype == np.float32 and bins.flags.c_contiguous and bins.size == x.shape[1] * height
    rows, n = x.shape
    w = np.ascontiguousarray(invert(window(n)))
    scale, offset = range_coeffs(range_min, range_max)
    product = np.empty((rows, n), np.complex64)
    spectrum = np.empty((rows, n), np.complex64)
    out = np.empty((rows, n), np.float32)
    fn = lib().jst_oracle_chain_pass
    fn.restype = None
    fn.argtypes = [C.c_void_p, _f32p, _f32p, C.c_uint64, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_uint64,
                   C.c_float, _f32p, _f32p, _f32p, _f32p]
    fft_ptr = C.cast(ref().ref_fft_c2c, C.c_void_p) if (use_ref and have_ref()) else None
    fn(fft_ptr, _p(x.view(np.float32)), _p(w.view(np.float32)), rows, n, amplitude_coeff(n), scale, offset, height,
       spectrogram_decay(rows), _p(product.view(np.float32)), _p(spectrum.view(np.float32)), _p(out), _p(bins))
    return out

# ------------------------------------------------------------- filter / FM side chains
def pad(x: np.ndarray, size: int, axis: int = -1) -> np.ndarray:
    """core/pad/module_impl_native_cpu.cc:75-140: zeros appended along axis."""
    shape = list(x.shape)
    shape[axis] = size
    return np.concatenate([x, np.zeros(shape, x.dtype)], axis=axis)


def unpad(x: np.ndarray, size: int, axis: int = -1):
    """core/unpad/module_impl_native_cpu.cc:66-135: (body, tail) split along axis."""
    n = x.shape[axis] - size
    body, tail = np.split(x, [n], axis=axis)
    return np.ascontiguousarray(body), np.ascontiguousarray(tail)


def fold(x: np.ndarray, axis: int, size: int, offset: int = 0, channel_axis=None,
         channel_offsets=None) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.complex64)
    oshape = list(x.shape)
    oshape[axis] = size
    out = np.empty(oshape, np.complex64)
    co = _u64(channel_offsets) if channel_offsets is not None else None
    lib().jst_oracle_fold_cf32(_p(x), _p(out), C.c_uint32(x.ndim), _p(_u64(x.shape), _u64p),
                               C.c_uint64(axis), C.c_uint64(offset), C.c_uint64(size),
                               C.c_int64(-1 if co is None else channel_axis),
                               _p(co, _u64p) if co is not None else None)
    return out


def arithmetic_add(x: np.ndarray, axis: int) -> np.ndarray:
    """Left-to-right F32 sum from +0 along axis (keepdims), components summed independently."""
    x = np.ascontiguousarray(x)
    axis %= x.ndim
    outer = int(np.prod(x.shape[:axis], dtype=np.uint64))
    r = x.shape[axis]
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.uint64))
    lanes = 2 if x.dtype == np.complex64 else 1
    flat = x.view(np.float32).reshape(outer, r, inner * lanes)
    out = np.empty((outer, inner * lanes), np.float32)
    lib().jst_oracle_arithmetic_add_f32(_p(np.ascontiguousarray(flat)), _p(out), C.c_uint64(outer),
                                        C.c_uint64(r), C.c_uint64(inner * lanes))
    oshape = list(x.shape)
    oshape[axis] = 1
    return out.view(x.dtype).reshape(oshape)


def filter_taps(sample_rate: float, bandwidth: float, center, taps: int) -> np.ndarray:
    center = np.ascontiguousarray(np.atleast_1d(center), dtype=np.float64)
    out = np.empty((center.size, taps), np.complex64)
    lib().jst_oracle_filter_taps(_p(out), C.c_double(sample_rate), C.c_double(bandwidth),
                                 center.ctypes.data_as(C.POINTER(C.c_double)),
                                 C.c_uint64(center.size), C.c_uint64(taps))
    return out


def overlap_add(buf: np.ndarray, ovl: np.ndarray, prev: np.ndarray, batch_axis=None):
    """dsp/overlap_add/module_impl_native_cpu.cc:121-202.  Returns (out, new_prev); the overlap
    region is the leading corner of the buffer (same coordinates)."""
    out = buf.copy()
    sl = tuple(slice(0, s) for s in ovl.shape)
    if batch_axis is None:
        out[sl] = out[sl] + prev
        return out, ovl.copy()
    shifted = np.concatenate([prev, np.delete(ovl, -1, axis=batch_axis)], axis=batch_axis)
    out[sl] = out[sl] + shifted
    return out, np.take(ovl, [ovl.shape[batch_axis] - 1], axis=batch_axis).copy()


def phase_correction(x: np.ndarray, increments, phases: np.ndarray, batch_axis=None,
                     channel_axis=None):
    """phases (F64[channels]) is updated in place, like the module's state."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    bc = x.shape[batch_axis] if batch_axis is not None else 1
    cc = x.shape[channel_axis] if channel_axis is not None else 1
    binner = int(np.prod(x.shape[batch_axis + 1:], dtype=np.uint64)) if batch_axis is not None else 1
    cinner = int(np.prod(x.shape[channel_axis + 1:], dtype=np.uint64)) if channel_axis is not None else 1
    inc = np.ascontiguousarray(np.broadcast_to(np.asarray(increments, np.float64), (cc,)))
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    lib().jst_oracle_phase_correction(_p(x), _p(out), C.c_uint64(x.size), C.c_uint64(bc),
                                      C.c_uint64(binner), C.c_uint64(cc), C.c_uint64(cinner),
                                      inc.ctypes.data_as(dp), phases.ctypes.data_as(dp))
    return out


class FmLane:
    """One lane of the FM demodulator with its carried state (oracle/jst_oracle.c fm_*)."""

    def __init__(self, mode="narrow", deemphasis="none", sample_rate=240e3):
        self._k = C.create_string_buffer(int(lib().jst_oracle_fm_coeffs_size()))
        self._st = C.create_string_buffer(int(lib().jst_oracle_fm_state_size()))
        self.wide = mode == "wide"
        lib().jst_oracle_fm_coeffs(self._k, C.c_int(self.wide),
                                   C.c_int({"none": 0, "50us": 1, "75us": 2}[deemphasis]),
                                   C.c_float(sample_rate))

    def __call__(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x.reshape(-1), dtype=np.complex64)
        out = np.empty(x.size * (2 if self.wide else 1), np.float32)
        lib().jst_oracle_fm_lane(_p(x), _p(out), C.c_uint64(x.size), self._k, self._st)
        return out.reshape(-1, 2) if self.wide else out


def filter_block(x: np.ndarray, plan: dict, sample_rate: float, bandwidth: float, center,
                 taps: int, state: dict) -> np.ndarray:
    """The Filter block's module chain on the CPU oracle (filter/block_impl.cc:350-582) for a
    signal [batch, samples] (batchAxis 0, sampleAxis 1) -> [batch, heads, out].  `state` carries
    the overlap and phase state across calls ({} to start)."""
    import math
    heads = len(center)
    b, s = x.shape
    conv = plan["convolutionSize"]
    sr, bw = float(np.float32(sample_rate)), float(np.float32(bandwidth))
    ctr = [float(np.float32(c)) for c in center]
    tapsv = filter_taps(sr, bw, ctr, taps)                          # [heads, taps]
    sig = pad(x.reshape(b, 1, s).astype(np.complex64), taps - 1, 2)  # [b, 1, conv]
    fsig = fft_c2c(sig, True)
    ffil = fft_c2c(pad(tapsv, s - 1, 1), True).reshape(1, heads, conv)
    prod = multiply(fsig, ffil)                                      # [b, heads, conv]
    spec = prod
    if plan["resample"]:
        spec = fold(prod, 2, plan["resamplerSize"], 0, 1, plan["resamplerOffsets"])
    time = fft_c2c(spec, False)
    c = np.float32(1.0) / np.float32(time.shape[2])
    norm = (time.real * c + 1j * (time.imag * c)).astype(np.complex64)
    if plan["resample"] and any(o != 0 for o in plan["resamplerOffsets"]):
        inc = [math.remainder(2.0 * math.pi * float(o) * float(s) / float(conv), 2.0 * math.pi)
               for o in plan["resamplerOffsets"]]
        phases = state.setdefault("phases", np.zeros(heads, np.float64))
        norm = phase_correction(norm, inc, phases, batch_axis=0, channel_axis=1)
    if plan["padSize"] == 0:
        return norm
    body, tail = unpad(norm, plan["padSize"], 2)
    prev = state.get("prev")
    if prev is None:
        prev = np.zeros((1,) + tail.shape[1:], np.complex64)
    out, state["prev"] = overlap_add(body, tail, prev, batch_axis=0)
    return out


def filter_engine_block(x: np.ndarray, taps: np.ndarray, plan: dict, state: dict) -> np.ndarray:
    """The filter_engine block's module chain (filter_engine/block_impl.cc:400-673) for a signal [samples] or
    [batch, samples] and EXTERNAL coefficients [T] (one head: no head axis, scalar fold offset / phase increment) or
    [C, T] (C heads: output [.., C, out]).  `plan` = the block's candidate plan (resample, offsets, sizes)."""
    import math
    x = np.asarray(x, np.complex64)
    rank1 = x.ndim == 1
    xb = x.reshape(1, -1) if rank1 else x
    b, s = xb.shape
    taps = np.asarray(taps, np.complex64)
    multi = taps.ndim == 2
    heads = taps.shape[0] if multi else 1
    tv = taps.reshape(heads, -1)
    t = tv.shape[1]
    conv = plan["convolutionSize"]
    assert conv == s + t - 1
    sig = pad(xb.reshape(b, 1, s), t - 1, 2)
    fsig = fft_c2c(sig, True)
    ffil = fft_c2c(pad(tv, s - 1, 1), True).reshape(1, heads, conv)
    spec = multiply(fsig, ffil)
    offsets = plan["resamplerOffsets"]
    if plan["resample"]:
        if multi:
            spec = fold(spec, 2, plan["resamplerSize"], 0, 1, offsets)
        else:
            spec = fold(spec, 2, plan["resamplerSize"], offsets[0])
    time = fft_c2c(spec, False)
    c = np.float32(1.0) / np.float32(time.shape[2])
    norm = (time.real * c + 1j * (time.imag * c)).astype(np.complex64)
    if plan["resample"] and any(o != 0 for o in offsets):
        inc = [math.remainder(2.0 * math.pi * float(o) * float(s) / float(conv), 2.0 * math.pi) for o in offsets]
        phases = state.setdefault("phases", np.zeros(heads, np.float64))
        norm = phase_correction(norm, inc, phases, batch_axis=0, channel_axis=1)
    if plan["padSize"] == 0:
        out = norm
    else:
        body, tail = unpad(norm, plan["padSize"], 2)
        prev = state.get("prev")
        if prev is None:
            prev = np.zeros((1,) + tail.shape[1:], np.complex64)
        out, state["prev"] = overlap_add(body, tail, prev, batch_axis=0)
    if not multi:
        out = out[:, 0, :]
    return out[0] if rank1 else out


def lineplot(avg: np.ndarray, x: np.ndarray, averaging: int = 1, decimation: int = 1) -> None:
    """In-place update of the averaged trace avg (F32[width // decimation]); x: F32 [batches, width]."""
    assert x.dtype == np.float32 and x.ndim == 2 and avg.dtype == np.float32
    lib().jst_oracle_lineplot(_p(avg), _p(x), C.c_uint64(x.shape[0]), C.c_uint64(avg.size),
                              C.c_uint64(x.strides[0] // 4), C.c_uint64(x.strides[1] // 4),
                              C.c_uint64(decimation), C.c_uint64(averaging))


def agc(x: np.ndarray, axis: int = -1, tile: int = 1024, reference: float = 1.0,
        epsilon: float = 1e-12, min_gain: float = 0.01, max_gain: float = 100.0,
        max_gain_change: float = 4.0) -> np.ndarray:
    """Tiled RMS AGC along `axis` (every other coordinate is an independent lane)."""
    assert x.dtype in (np.float32, np.complex64)
    moved = np.ascontiguousarray(np.moveaxis(x, axis, -1))
    out = np.empty_like(moved)
    samples = moved.shape[-1]
    lanes = moved.size // samples
    lib().jst_oracle_agc(_p(moved.view(np.float32)), _p(out.view(np.float32)),
                         C.c_int(1 if x.dtype == np.complex64 else 0), C.c_uint64(lanes),
                         C.c_uint64(samples), C.c_uint64(tile), C.c_double(reference),
                         C.c_double(epsilon), C.c_double(min_gain), C.c_double(max_gain),
                         C.c_double(max_gain_change))
    return np.moveaxis(out, -1, axis)


_CAST_SCALER = {np.dtype(np.int8): 128.0, np.dtype(np.uint8): 128.0, np.dtype(np.int16): 32768.0,
                np.dtype(np.uint16): 32768.0, np.dtype(np.int32): 2147483648.0,
                np.dtype(np.uint32): 2147483648.0}


def cast(x: np.ndarray, complex_pairs: bool = False) -> np.ndarray:
    """core/cast/module_impl_native_cpu.cc:137-283: F32(in) / scaler (128, 32768, 2^31);
    complex_pairs: the last axis holds (re, im) -> CF32.  F32 input -> CF32 with imag 0."""
    if x.dtype == np.float32:
        return x.astype(np.complex64)
    s = np.float32(_CAST_SCALER[x.dtype])
    y = x.astype(np.float32) / s  # int -> F32 rounds to nearest even, the division is exact
    if complex_pairs:
        return np.ascontiguousarray(y).view(np.complex64)[..., 0]
    return y


def add(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """core/add/module_impl_native_cpu.cc:83-98 (broadcasting like multiply)."""
    assert a.dtype == b.dtype and a.dtype in (np.float32, np.complex64)
    return (a + b).astype(a.dtype)


def squelch(x: np.ndarray, threshold: float):
    """dsp/squelch/module_impl_native_cpu.cc:66-98: (passing, peak) with peak = max |x| in F32 -- std::max
    keeps the running peak when a value is NaN; |z| of a complex sample is libm hypotf =
    (float)sqrt((double)re*re + (double)im*im), infinite as soon as a part is."""
    x = np.asarray(x)
    if np.iscomplexobj(x):
        z = x.astype(np.complex64).reshape(-1)
        re, im = z.real.astype(np.float64), z.imag.astype(np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            mag = np.sqrt(re * re + im * im).astype(np.float32)
        mag[np.isinf(re) | np.isinf(im)] = np.inf
    else:
        mag = np.abs(x.astype(np.float32).reshape(-1))
    peak = np.float32(0.0)
    finite_or_inf = mag[~np.isnan(mag)]
    if finite_or_inf.size:
        peak = np.float32(max(peak, finite_or_inf.max()))
    return bool(peak > np.float32(threshold)), peak


class AmLane:
    """One lane of the AM demodulator (dsp/am/module_impl_native_cpu.cc:40-101): envelope = |x| (libm hypotf
    like `squelch`), y[n] = (e[n] - e[n-1]) + alpha * y[n-1] in F32, state carried across calls."""

    def __init__(self, dc_alpha: float = 0.995):
        self.alpha = np.float32(dc_alpha)
        self.prev_env = np.float32(0.0)
        self.prev_out = np.float32(0.0)

    def __call__(self, x: np.ndarray) -> np.ndarray:
        z = np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.complex64)
        re, im = z.real.astype(np.float64), z.imag.astype(np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            env = np.sqrt(re * re + im * im).astype(np.float32)
            env[np.isinf(re) | np.isinf(im)] = np.inf
            diff = env - np.concatenate(([self.prev_env], env[:-1])).astype(np.float32)  # one F32 rounding each
            out = np.empty(z.size, np.float32)
            y, a = self.prev_out, self.alpha
            for n in range(z.size):
                y = np.float32(diff[n] + np.float32(a * y))
                out[n] = y
        if z.size:
            self.prev_env, self.prev_out = env[-1], y
        return out
