"""Python host mirror of the Jetstream module interface over the C ABI (include/jetstream_hip.h).

Names follow the reference: ``Tensor`` (include/jetstream/memory/tensor.hh), ``Module`` built
through the registry with ``(type, device, runtime, provider)`` (include/jetstream/registry.hh:
119-125), ``Runtime`` (include/jetstream/runtime.hh:22-40) and the ``SpectrumEngine`` block
wiring (src/domains/dsp/spectrum_engine/block_impl.cc:120-217).  Every call goes through
``libjetstream_hip.so``; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libjetstream_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built "
        "(run `make -C cyberether_amd/csrc`). There is no CPU fallback."
    )

# PyTorch-ROCm bundles its own libamdhip64; whichever copy is loaded FIRST serves the whole process.
# Loading torch's before ours keeps one HIP runtime in the process, so torch tensors / streams /
# torch.distributed (RCCL) and this library see the same device context.
try:  # plumbing only -- nothing below needs torch
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_lib = C.CDLL(LIB_PATH)

MAX_RANK = 8
DEVICE = {"none": 1, "cpu": 2, "hip": 64}
DEVICE_NAME = {v: k for k, v in DEVICE.items()}
DTYPE = {"F32": 1, "CF32": 2, "F64": 3, "U64": 4, "I8": 5, "CI8": 6, "I16": 7, "CI16": 8, "U8": 9,
         "CU8": 10, "U16": 11, "CU16": 12, "I32": 13, "CI32": 14, "U32": 15, "CU32": 16,
         "CF64": 17}
# complex integer formats are interleaved (re, im) pairs: numpy sees them as a trailing axis of 2
_CINT = {6: np.int8, 8: np.int16, 10: np.uint8, 12: np.uint16, 14: np.int32, 16: np.uint32}
NP_DTYPE = {1: np.float32, 2: np.complex64, 3: np.float64, 4: np.uint64, 5: np.int8, 7: np.int16,
            9: np.uint8, 11: np.uint16, 13: np.int32, 15: np.uint32, 17: np.complex128}
DTYPE_OF_NP = {np.dtype(v): k for k, v in NP_DTYPE.items()}
RESULT_NAMES = ["SUCCESS", "ERROR", "WARNING", "FATAL", "SKIP", "YIELD", "RELOAD", "RECREATE",
                "TIMEOUT", "INCOMPLETE"]

RUNTIME_GRAPH, RUNTIME_FUSE, RUNTIME_TIMING, RUNTIME_PIPELINE, RUNTIME_COMBINE, RUNTIME_BATCH = 1, 2, 4, 8, 16, 32

TAINT = {"IN_PLACE": 1, "DISCONTIGUOUS": 2, "SURFACE": 4, "CROSS_DEVICE": 16,
         "STATIC_OUTPUT": 64, "STATELESS": 128}


class JetstreamError(RuntimeError):
    """A non-SUCCESS Result crossed the ABI; .result is the reference Result name."""

    def __init__(self, result: int, message: str):
        self.result = RESULT_NAMES[result] if result < len(RESULT_NAMES) else str(result)
        super().__init__(f"{self.result}: {message}")


class _Desc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offset", C.c_uint64), ("dtype", C.c_uint8),
                ("device", C.c_uint8), ("rank", C.c_uint32), ("shape", C.c_uint64 * MAX_RANK),
                ("stride", C.c_uint64 * MAX_RANK), ("sample_axis", C.c_int64),
                ("batch_axis", C.c_int64), ("channel_axis", C.c_int64)]


def _sig(name, restype, *argtypes):
    fn = getattr(_lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_u64p = C.POINTER(C.c_uint64)
_h = C.c_void_p
_hp = C.POINTER(C.c_void_p)
_strs = C.POINTER(C.c_char_p)
R = C.c_uint16

_sig("jst_version", C.c_char_p)
_sig("jst_debug_set", R, C.c_char_p, C.c_char_p)
_sig("jst_last_error", C.c_char_p)
_sig("jst_device_count", C.c_int)
_sig("jst_device_set", R, C.c_int)
_sig("jst_registry_list", C.c_size_t, C.c_char_p, C.c_size_t)
_sig("jst_tensor_create", R, C.c_uint8, C.c_uint8, C.c_uint32, _u64p, _hp)
_sig("jst_tensor_create_ring", R, C.c_uint8, C.c_uint8, C.c_uint32, _u64p, C.c_uint64, _hp)
_sig("jst_tensor_wrap", R, C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint8, C.c_uint32, _u64p, _u64p,
     C.c_uint64, _hp)
_sig("jst_tensor_rebind", R, _h, C.c_void_p, C.c_size_t)
_sig("jst_tensor_clone", R, _h, _hp)
_sig("jst_tensor_view", R, _h, C.c_uint32, _u64p, _u64p, C.c_uint64, _hp)
_sig("jst_tensor_copy", R, _h, _h, C.c_void_p)
_sig("jst_tensor_destroy", R, _h)
_sig("jst_tensor_describe", R, _h, C.POINTER(_Desc))
_sig("jst_tensor_ring_select", R, _h, C.c_uint64)
_sig("jst_tensor_reshape", R, _h, C.c_uint32, _u64p)
_sig("jst_tensor_expand_dims", R, _h, C.c_uint64)
_sig("jst_tensor_squeeze_dims", R, _h, C.c_uint64)
_sig("jst_tensor_slice", R, _h, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64)
_sig("jst_tensor_permute", R, _h, C.c_uint32, _u64p)
_sig("jst_tensor_broadcast_to", R, _h, C.c_uint32, _u64p)
_sig("jst_tensor_set_attribute_u64", R, _h, C.c_char_p, C.c_uint64)
_sig("jst_tensor_set_attribute_f64", R, _h, C.c_char_p, C.c_double)
_sig("jst_tensor_set_attribute_u64v", R, _h, C.c_char_p, _u64p, C.c_uint64)
_sig("jst_tensor_set_attribute_f64v", R, _h, C.c_char_p, C.POINTER(C.c_double), C.c_uint64)
_sig("jst_tensor_remove_attribute", R, _h, C.c_char_p)
_sig("jst_tensor_get_attribute_f64v", R, _h, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64))
_sig("jst_tensor_copy_from_host", R, _h, C.c_void_p, C.c_size_t)
_sig("jst_tensor_copy_to_host", R, _h, C.c_void_p, C.c_size_t)
_sig("jst_tensor_copy_from_host_async", R, _h, C.c_void_p, C.c_size_t)
_sig("jst_module_create", R, C.c_char_p, C.c_uint8, C.c_char_p, C.c_char_p, _strs, C.c_uint32,
     _strs, _hp, C.c_uint32, _hp)
_sig("jst_module_destroy", R, _h)
_sig("jst_module_reconfigure", R, _h, _strs, C.c_uint32, C.c_int)
_sig("jst_module_output", R, _h, C.c_char_p, _hp)
_sig("jst_module_state", R, _h, C.c_char_p, _hp)
_sig("jst_module_taint", C.c_uint64, _h)
_sig("jst_module_timing", R, _h, _u64p, C.POINTER(C.c_double))
_sig("jst_module_compute_initialize", R, _h)
_sig("jst_module_compute_submit", R, _h, C.c_void_p)
_sig("jst_module_compute_deinitialize", R, _h)
_sig("jst_runtime_create", R, _hp, C.c_uint32, C.c_uint32, _hp)
_sig("jst_runtime_destroy", R, _h)
_sig("jst_runtime_compute", R, _h, C.c_uint64, C.c_int)
_sig("jst_runtime_synchronize", R, _h)
_sig("jst_runtime_stream", C.c_void_p, _h)
_sig("jst_runtime_period", C.c_uint64, _h)
_sig("jst_runtime_graph_active", C.c_int, _h)
_sig("jst_runtime_order", C.c_size_t, _h, C.c_char_p, C.c_size_t)
_sig("jst_ring_push", R, _h, C.c_void_p, C.c_uint64)
_sig("jst_probe_ring_push_chunks", R, _h, C.c_void_p, C.c_uint64, C.c_uint64)
_sig("jst_ring_acquire", R, _h, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64))
_sig("jst_ring_commit", R, _h, C.c_uint64)
_sig("jst_ring_wait", R, _h, C.c_uint64, C.c_uint32)
_sig("jst_ring_clear", R, _h)
_sig("jst_ring_size", C.c_uint64, _h)
_sig("jst_ring_capacity", C.c_uint64, _h)
_sig("jst_ring_overflows", C.c_uint64, _h)
_sig("jst_runtime_units", C.c_size_t, _h, C.c_char_p, C.c_size_t)
_sig("jst_runtime_unit_mean_ms", C.c_double, _h, C.c_char_p)
_sig("jst_runtime_unit_mean_cycles", C.c_double, _h, C.c_char_p)
_sig("jst_runtime_batched", C.c_int, _h)
_sig("jst_runtime_branches", C.c_int, _h)
_sig("jst_runtime_event_overhead_ms", C.c_double, _h)
_sig("jst_runtime_reset_timing", R, _h)
_sig("jst_comm_available", C.c_int)
_sig("jst_comm_unique_id", R, C.c_char_p)
_sig("jst_comm_init", R, C.c_uint32, C.c_uint32, C.c_char_p, _hp)
_sig("jst_comm_destroy", R, _h)
_sig("jst_comm_rank", C.c_uint32, _h)
_sig("jst_comm_world", C.c_uint32, _h)
_sig("jst_comm_calls", C.c_uint64, _h)
_sig("jst_comm_uses_rccl", C.c_int, _h)
_sig("jst_comm_allreduce", R, _h, _h, C.c_int, C.c_int, C.c_void_p)
_sig("jst_fft_twiddles", R, C.c_uint64, C.POINTER(C.c_float))
_sig("jst_probe_fft_path", C.c_int, C.c_uint64)
_sig("jst_probe_tanhf", R, C.c_void_p, C.c_void_p, C.c_uint64)
_sig("jst_probe_exact_sweep", R, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint64),
     C.POINTER(C.c_uint64), C.POINTER(C.c_uint32))
_sig("jst_probe_amplitude_range", R, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float,
     C.c_float, C.c_float, C.c_float)


def _check(result: int) -> None:
    if result not in (0, 6):  # SUCCESS | RELOAD
        raise JetstreamError(result, _lib.jst_last_error().decode(errors="replace"))


def _arr(values: Sequence[int]):
    return (C.c_uint64 * max(len(values), 1))(*[int(v) for v in values])


def debug_set(name: str, value: Optional[str]) -> None:
    """jst_debug_set: flip one of the library's A/B switches (JST_FFT_KERNEL, JST_QUAD_STATIC, JST_FM_SERIAL, JST_RUNTIME_*) for
    this process; None unsets it.  The environment variable of the same name only seeds the switch when the library first asks."""
    _check(_lib.jst_debug_set(name.encode(), None if value is None else str(value).encode()))


def version() -> str:
    return _lib.jst_version().decode()


def device_count() -> int:
    return int(_lib.jst_device_count())


def set_device(ordinal: int) -> None:
    _check(_lib.jst_device_set(ordinal))


def list_available_modules() -> List[str]:
    buf = C.create_string_buffer(1 << 16)
    _lib.jst_registry_list(buf, len(buf))
    return [l for l in buf.value.decode().split("\n") if l]


def fft_twiddles(n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.complex64)
    _check(_lib.jst_fft_twiddles(n, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


FFT_PATHS = ("register", "tile", "tile_pair", "passes")


def fft_path(n: int) -> str:
    """Kernel family a complex transform of (pass) length n runs on: JST_FFT_PATH_* of include/jetstream_hip.h."""
    return FFT_PATHS[_lib.jst_probe_fft_path(n)]


class Tensor:
    """Handle to a Jetstream tensor (shape/stride/offset in elements, shared storage)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:  # module globals are already gone at interpreter shutdown
            _lib.jst_tensor_destroy(h)

    # -- construction ---------------------------------------------------------------------
    @staticmethod
    def create(device: str, dtype: str, shape: Sequence[int], slots: int = 1) -> "Tensor":
        out = C.c_void_p()
        _check(_lib.jst_tensor_create_ring(DEVICE[device], DTYPE[dtype], len(shape), _arr(shape),
                                           slots, C.byref(out)))
        return Tensor(out.value)

    @staticmethod
    def from_numpy(array: np.ndarray, device: str = "hip", dtype: Optional[str] = None,
                   **axes) -> "Tensor":
        """dtype: only needed for the complex integer sample formats ("CI8", "CI16", ...), which
        numpy has no type for: pass an integer array whose last axis holds (re, im)."""
        array = np.ascontiguousarray(array)
        if dtype is not None and DTYPE[dtype] in _CINT:
            t = Tensor.create(device, dtype, array.shape[:-1])
        else:
            t = Tensor.create(device, {v: k for k, v in DTYPE.items()}[DTYPE_OF_NP[array.dtype]],
                              array.shape)
        t.copy_from(array)
        if axes:
            t.set_axes(**axes)
        return t

    @staticmethod
    def wrap(ptr: int, nbytes: int, device: str, dtype: str, shape: Sequence[int],
             stride: Optional[Sequence[int]] = None, offset: int = 0) -> "Tensor":
        """Borrow external memory (e.g. ``torch_tensor.data_ptr()``); the caller keeps it alive."""
        out = C.c_void_p()
        _check(_lib.jst_tensor_wrap(C.c_void_p(ptr), nbytes, DEVICE[device], DTYPE[dtype],
                                    len(shape), _arr(shape), _arr(stride) if stride else None,
                                    offset, C.byref(out)))
        return Tensor(out.value)

    def clone(self) -> "Tensor":
        out = C.c_void_p()
        _check(_lib.jst_tensor_clone(self._h, C.byref(out)))
        return Tensor(out.value)

    def view(self, shape: Sequence[int], stride: Optional[Sequence[int]] = None, offset: int = 0) -> "Tensor":
        """jst_tensor_view: a new handle on this tensor's STORAGE with the caller's geometry (elements; stride None =
        dense) -- storage identity, and with it the runtime's data-flow edges, is kept (unlike wrap)."""
        out = C.c_void_p()
        _check(_lib.jst_tensor_view(self._h, len(shape), _arr(shape), _arr(stride) if stride else None, offset,
                                    C.byref(out)))
        return Tensor(out.value)

    def rebind(self, ptr: int, nbytes: int) -> "Tensor":
        """jst_tensor_rebind: move this tensor's storage (and every view of it, e.g. a module's output) onto external
        memory the caller owns -- a host framework that allocated the buffer itself."""
        _check(_lib.jst_tensor_rebind(self._h, C.c_void_p(ptr), nbytes))
        return self

    # -- introspection --------------------------------------------------------------------
    def _desc(self) -> _Desc:
        d = _Desc()
        _check(_lib.jst_tensor_describe(self._h, C.byref(d)))
        return d

    @property
    def shape(self):
        d = self._desc()
        return tuple(int(d.shape[i]) for i in range(d.rank))

    @property
    def stride(self):
        d = self._desc()
        return tuple(int(d.stride[i]) for i in range(d.rank))

    @property
    def offset(self) -> int:
        return int(self._desc().offset)

    @property
    def dtype(self) -> str:
        return {v: k for k, v in DTYPE.items()}[self._desc().dtype]

    @property
    def device(self) -> str:
        return DEVICE_NAME[self._desc().device]

    @property
    def data_ptr(self) -> int:
        return int(self._desc().data or 0)

    @property
    def axes(self) -> Dict[str, Optional[int]]:
        d = self._desc()
        f = lambda v: None if v < 0 else int(v)
        return {"sample": f(d.sample_axis), "batch": f(d.batch_axis), "channel": f(d.channel_axis)}

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.uint64)) if self.shape else 0

    # -- views (mutating, like the reference) -----------------------------------------------
    def reshape(self, shape):
        _check(_lib.jst_tensor_reshape(self._h, len(shape), _arr(shape)))
        return self

    def expand_dims(self, axis):
        _check(_lib.jst_tensor_expand_dims(self._h, axis))
        return self

    def squeeze_dims(self, axis):
        _check(_lib.jst_tensor_squeeze_dims(self._h, axis))
        return self

    def slice(self, axis, begin, end, step=1):
        _check(_lib.jst_tensor_slice(self._h, axis, begin, end, step))
        return self

    def permute(self, axes):
        _check(_lib.jst_tensor_permute(self._h, len(axes), _arr(axes)))
        return self

    def broadcast_to(self, shape):
        _check(_lib.jst_tensor_broadcast_to(self._h, len(shape), _arr(shape)))
        return self

    def ring_select(self, slot: int):
        _check(_lib.jst_tensor_ring_select(self._h, slot))
        return self

    # -- attributes -----------------------------------------------------------------------
    def set_axes(self, sample=None, batch=None, channel=None):
        for key, v in (("sampleAxis", sample), ("batchAxis", batch), ("channelAxis", channel)):
            if v is None:
                _check(_lib.jst_tensor_remove_attribute(self._h, key.encode()))
            else:
                _check(_lib.jst_tensor_set_attribute_u64(self._h, key.encode(), int(v)))
        return self

    def set_attribute(self, key: str, value):
        if isinstance(value, (list, tuple, np.ndarray)):
            if all(isinstance(v, (int, np.integer)) for v in value):
                arr = (C.c_uint64 * len(value))(*[int(v) for v in value])
                _check(_lib.jst_tensor_set_attribute_u64v(self._h, key.encode(), arr, len(value)))
            else:
                arr = (C.c_double * len(value))(*[float(v) for v in value])
                _check(_lib.jst_tensor_set_attribute_f64v(self._h, key.encode(), arr, len(value)))
            return self
        if isinstance(value, (int, np.integer)):
            _check(_lib.jst_tensor_set_attribute_u64(self._h, key.encode(), int(value)))
        else:
            _check(_lib.jst_tensor_set_attribute_f64(self._h, key.encode(), float(value)))
        return self

    # -- data movement (dense only) -------------------------------------------------------------
    def attribute(self, key: str):
        """Scalar attributes come back as float, vector attributes as a list; raises when absent."""
        n = C.c_uint64(0)
        _check(_lib.jst_tensor_get_attribute_f64v(self._h, key.encode(), None, C.byref(n)))
        buf = (C.c_double * max(n.value, 1))()
        _check(_lib.jst_tensor_get_attribute_f64v(self._h, key.encode(), buf, C.byref(n)))
        vals = [float(buf[i]) for i in range(n.value)]
        return vals[0] if n.value == 1 else vals

    def copy_from(self, array: np.ndarray, asynchronous: bool = False):
        code = self._desc().dtype
        if code in _CINT:  # interleaved (re, im) integer pairs: array[..., 2]
            array = np.ascontiguousarray(array, dtype=_CINT[code])
            if tuple(array.shape) != tuple(self.shape) + (2,):
                raise ValueError("complex integer tensors take an array with a trailing axis of 2")
        else:
            array = np.ascontiguousarray(array, dtype=NP_DTYPE[code])
        fn = _lib.jst_tensor_copy_from_host_async if asynchronous else _lib.jst_tensor_copy_from_host
        _check(fn(self._h, array.ctypes.data_as(C.c_void_p), array.nbytes))
        return self

    def copy_from_tensor(self, source: "Tensor", stream: int = 0):
        """jst_tensor_copy: dense device-to-device copy of `source` into this tensor, enqueued on `stream`."""
        _check(_lib.jst_tensor_copy(self._h, source._h, C.c_void_p(stream) if stream else None))
        return self

    def numpy(self) -> np.ndarray:
        code = self._desc().dtype
        if code in _CINT:
            out = np.empty(tuple(self.shape) + (2,), dtype=_CINT[code])
        else:
            out = np.empty(self.shape, dtype=NP_DTYPE[code])
        _check(_lib.jst_tensor_copy_to_host(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out


class Module:
    """Registry::BuildModule + Module::create in one step (src/module.cc:47-212)."""

    def __init__(self, type: str, config: Optional[dict] = None,
                 inputs: Optional[Dict[str, Tensor]] = None, name: Optional[str] = None,
                 device: str = "hip", provider: str = "generic"):
        self.type = type
        self.name = name or type
        self._inputs = dict(inputs or {})  # keep the tensor handles alive
        cfg = [f"{k}={_cfg_value(v)}".encode() for k, v in (config or {}).items()]
        ports = [k.encode() for k in self._inputs]
        handles = [t._h for t in self._inputs.values()]
        out = C.c_void_p()
        cfg_arr = (C.c_char_p * max(len(cfg), 1))(*cfg)
        port_arr = (C.c_char_p * max(len(ports), 1))(*ports)
        h_arr = (C.c_void_p * max(len(handles), 1))(*handles)
        self._h = None
        _check(_lib.jst_module_create(type.encode(), DEVICE[device], provider.encode(),
                                      self.name.encode(), cfg_arr, len(cfg), port_arr, h_arr,
                                      len(ports), C.byref(out)))
        self._h = C.c_void_p(out.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.jst_module_destroy(h)

    def reconfigure(self, config: dict, validate_only: bool = False) -> str:
        """Module::reconfigure (src/module.cc:233-290): "success" when the change was applied in place (or
        nothing changed), "recreate" when the module must be rebuilt for it (nothing was changed); an
        invalid configuration raises and leaves the module as it was.  A Runtime holding the module
        re-captures its graph on the next compute()."""
        cfg = [f"{k}={_cfg_value(v)}".encode() for k, v in config.items()]
        arr = (C.c_char_p * max(len(cfg), 1))(*cfg)
        r = _lib.jst_module_reconfigure(self._h, arr, len(cfg), 1 if validate_only else 0)
        if r == 7:
            return "recreate"
        _check(r)
        return "success"

    def output(self, port: str) -> Tensor:
        out = C.c_void_p()
        _check(_lib.jst_module_output(self._h, port.encode(), C.byref(out)))
        return Tensor(out.value)

    def state(self, key: str) -> Tensor:
        out = C.c_void_p()
        _check(_lib.jst_module_state(self._h, key.encode(), C.byref(out)))
        return Tensor(out.value)

    @property
    def taint(self) -> int:
        return int(_lib.jst_module_taint(self._h))

    # -- producer side of a live ring_source (circular_buffer.hh:31-48 against the HBM ring) ---------
    def ring_push(self, samples: np.ndarray) -> str:
        """push(): any number of elements (complex samples of the source's dtype; integer formats as [..., 2]
        arrays).  Returns "success" or "incomplete" (overflow policy reject: nothing was taken)."""
        a, count = self._ring_samples(samples)
        r = _lib.jst_ring_push(self._h, a.ctypes.data_as(C.c_void_p), count)
        if r == 9:
            return "incomplete"
        _check(r)
        return "success"

    def _ring_samples(self, samples):
        """(contiguous array, element count) after checking the array against the SOURCE's sample format: the library
        reads `count` elements of that format from the buffer (an int8 array pushed into a CF32 source would be read four
        times past its end).  The format is looked up once per module."""
        a = np.ascontiguousarray(samples)
        want = getattr(self, "_ring_format", None)
        if want is None:
            fmt = self.output("buffer").dtype
            want = {"CF32": (np.dtype(np.complex64), 8, False, fmt), "CI16": (np.dtype(np.int16), 4, True, fmt),
                    "CI8": (np.dtype(np.int8), 2, True, fmt), "CU8": (np.dtype(np.uint8), 2, True, fmt)}.get(fmt)
            if want is None:
                raise JetstreamError(1, f"[MODULE_RING_SOURCE] ring_push: unsupported source format {fmt}")
            self._ring_format = want
        if a.dtype != want[0] or (want[2] and (a.ndim == 0 or a.shape[-1] != 2)):
            raise JetstreamError(1, f"[MODULE_RING_SOURCE] ring_push: a {want[3]} source takes "
                                    f"{'[..., 2] ' if want[2] else ''}{want[0]} samples, got {a.dtype} {a.shape}")
        return a, a.nbytes // want[1]

    def ring_push_chunks(self, samples: np.ndarray, chunk: int = 8192) -> str:
        """The reference's producer loop in native code (jst_probe_ring_push_chunks): consecutive pushes of <= chunk
        elements, as the Soapy thread does (soapy/module_impl.cc:375-399)."""
        a, count = self._ring_samples(samples)
        r = _lib.jst_probe_ring_push_chunks(self._h, a.ctypes.data_as(C.c_void_p), count, chunk)
        if r == 9:
            return "incomplete"
        _check(r)
        return "success"

    def ring_acquire(self):
        """(address, max_elements) of the pinned staging memory the next samples go to (zero-copy producer)."""
        ptr, room = C.c_void_p(), C.c_uint64()
        _check(_lib.jst_ring_acquire(self._h, C.byref(ptr), C.byref(room)))
        return int(ptr.value), int(room.value)

    def ring_commit(self, count: int) -> str:
        r = _lib.jst_ring_commit(self._h, count)
        if r == 9:
            return "incomplete"
        _check(r)
        return "success"

    def ring_wait(self, size: int, timeout_ms: int = 5000) -> bool:
        """waitForSize(): True once `size` elements are buffered, False on timeout."""
        r = _lib.jst_ring_wait(self._h, size, timeout_ms)
        if r == 8:
            return False
        _check(r)
        return True

    def ring_clear(self):
        _check(_lib.jst_ring_clear(self._h))

    @property
    def ring_size(self) -> int:
        return int(_lib.jst_ring_size(self._h))

    @property
    def ring_capacity(self) -> int:
        return int(_lib.jst_ring_capacity(self._h))

    @property
    def ring_overflows(self) -> int:
        return int(_lib.jst_ring_overflows(self._h))

    @property
    def timing(self):
        cycles, ms = C.c_uint64(), C.c_double()
        _check(_lib.jst_module_timing(self._h, C.byref(cycles), C.byref(ms)))
        return {"cycles": cycles.value, "computeTime": ms.value}


def _cfg_value(v) -> str:
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, float):
        return repr(v)
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(repr(float(x)) if isinstance(x, float) else str(int(x)) for x in v) + "]"
    return str(v)


class Runtime:
    """One device segment: ordered modules on one HIP stream, optionally as a hipGraph."""

    def __init__(self, modules: Iterable[Module], graph: bool = False, fuse: bool = False,
                 timing: bool = False, pipeline: bool = False, combine: bool = False, batch: Optional[bool] = None):
        # batch=None: cycle batching whenever it can apply (graph + fuse, no per-unit timing; the planner keeps any chain
        # that is not a resident ring -> spectrum unit -> span-capable readers per cycle, Runtime.batched tells).  What is
        # visible after compute() is bit-identical either way; pass batch=False for one launch per unit and cycle.
        if batch is None:
            batch = bool(graph and fuse and not timing)
        self.modules = list(modules)
        flags = (RUNTIME_GRAPH if graph else 0) | (RUNTIME_FUSE if fuse else 0) | \
                (RUNTIME_TIMING if timing else 0) | (RUNTIME_PIPELINE if pipeline else 0) | \
                (RUNTIME_COMBINE if combine else 0) | (RUNTIME_BATCH if batch else 0)
        arr = (C.c_void_p * max(len(self.modules), 1))(*[m._h for m in self.modules])
        out = C.c_void_p()
        self._h = None
        _check(_lib.jst_runtime_create(arr, len(self.modules), flags, C.byref(out)))
        self._h = C.c_void_p(out.value)

    def __del__(self):
        self.destroy()

    def destroy(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.jst_runtime_destroy(h)

    def compute(self, cycles: int = 1, sync: bool = True) -> str:
        """Runs `cycles` compute cycles.  "success", or -- the reference runtime's quiet endings
        (src/runtime/native/cpu/impl.cc:121-124) -- "yield" / "timeout" when a source had no data: the
        cycle in which that happened, and the rest of the request, did not run."""
        r = _lib.jst_runtime_compute(self._h, cycles, 1 if sync else 0)
        if r == 5:
            return "yield"
        if r == 8:
            return "timeout"
        _check(r)
        return "success"

    def synchronize(self):
        _check(_lib.jst_runtime_synchronize(self._h))

    @property
    def stream(self) -> int:
        return int(_lib.jst_runtime_stream(self._h) or 0)

    @property
    def period(self) -> int:
        return int(_lib.jst_runtime_period(self._h))

    @property
    def graph_active(self) -> bool:
        return bool(_lib.jst_runtime_graph_active(self._h))

    def _lines(self, fn) -> List[str]:
        buf = C.create_string_buffer(1 << 14)
        fn(self._h, buf, len(buf))
        return [l for l in buf.value.decode().split("\n") if l]

    @property
    def order(self) -> List[str]:
        return self._lines(_lib.jst_runtime_order)

    @property
    def units(self) -> List[str]:
        return self._lines(_lib.jst_runtime_units)

    @property
    def branches(self) -> int:
        """Parallel branches of a captured cycle (1 = one serial chain)."""
        return int(_lib.jst_runtime_branches(self._h))

    def unit_mean_ms(self, prefix: str) -> float:
        return float(_lib.jst_runtime_unit_mean_ms(self._h, prefix.encode()))

    def unit_mean_cycles(self, prefix: str) -> float:
        """Compute cycles one timing sample of the unit covers (1, or the ring period when cycle-batched)."""
        return float(_lib.jst_runtime_unit_mean_cycles(self._h, prefix.encode()))

    @property
    def batched(self) -> bool:
        """True when `batch=True` took effect: the cycles of a ring period run as one launch per unit."""
        return bool(_lib.jst_runtime_batched(self._h))

    def event_overhead_ms(self) -> float:
        return float(_lib.jst_runtime_event_overhead_ms(self._h))

    def reset_timing(self):
        _check(_lib.jst_runtime_reset_timing(self._h))


COMM_ID_BYTES = 128


def comm_available() -> bool:
    """Can RCCL be loaded in this process (jst_comm_available)?"""
    return bool(_lib.jst_comm_available())


def comm_unique_id() -> bytes:
    """ncclUniqueId of a new communicator (rank 0 calls this and ships the 128 bytes to the other ranks)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(_lib.jst_comm_unique_id(buf))
    return buf.raw


class Comm:
    """The path's communicator behind the C ABI (csrc/jst/comm.cc): RCCL over xGMI, one per process; world == 1 needs
    no RCCL.  all_reduce runs in place on a dense F32 / U32 HIP tensor, on `stream` (default: the null stream)."""

    def __init__(self, rank: int = 0, world: int = 1, unique_id: Optional[bytes] = None):
        out = C.c_void_p()
        self._h = None
        if unique_id is not None and len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        _check(_lib.jst_comm_init(rank, world, unique_id, C.byref(out)))
        self._h = C.c_void_p(out.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.jst_comm_destroy(h)

    rank = property(lambda self: int(_lib.jst_comm_rank(self._h)))
    world = property(lambda self: int(_lib.jst_comm_world(self._h)))
    calls = property(lambda self: int(_lib.jst_comm_calls(self._h)))
    uses_rccl = property(lambda self: bool(_lib.jst_comm_uses_rccl(self._h)))

    def all_reduce(self, tensor: "Tensor", op: str = "sum", average: bool = False, stream: int = 0) -> None:
        _check(_lib.jst_comm_allreduce(self._h, tensor._h, {"sum": 0, "max": 1}[op], 1 if average else 0,
                                       C.c_void_p(stream) if stream else None))


class SpectrumEngine:
    """The spectrum_engine BLOCK: cast -> window -> invert -> reshape -> multiply -> fft ->
    [agc] -> amplitude -> [range], wired exactly as
    src/domains/dsp/spectrum_engine/block_impl.cc:120-217 (enableAgc defaults to false,
    spectrum_engine/block.hh:9; with it the chain is not fused: agc sits between fft and amplitude)."""

    def __init__(self, buffer: Tensor, enable_scale: bool = True, range_min: float = -100.0,
                 range_max: float = 0.0, name: str = "spectrum", provider: str = "generic",
                 enable_agc: bool = False):
        """provider: registry key of the amplitude/range implementations -- "generic" restates the
        reference's libm arithmetic bit for bit, "fast" uses the hardware transcendentals."""
        axes = buffer.axes
        rank = len(buffer.shape)
        axis = axes["sample"] if axes["sample"] is not None else (0 if rank == 1 else None)
        if axis is None:
            raise JetstreamError(1, "[BLOCK_SPECTRUM_ENGINE] Input validation plan is unavailable.")
        n = buffer.shape[axis]
        p = name + "."
        self.cast = Module("cast", {"outputType": "CF32"}, {"buffer": buffer}, p + "cast_input")
        complex_input = self.cast.output("buffer")
        self.window = Module("window", {"size": n}, {}, p + "window")
        window_out = self.window.output("window").set_axes(sample=0)
        self.invert = Module("invert", {}, {"signal": window_out}, p + "invert")
        shape = [n if d == axis else 1 for d in range(rank)]
        self.reshape = Module("reshape", {"shape": shape},
                              {"buffer": self.invert.output("signal")}, p + "reshape_window")
        reshaped = self.reshape.output("buffer").set_axes(sample=axis)
        self.multiply = Module("multiply", {}, {"a": complex_input, "b": reshaped}, p + "multiply")
        self.fft = Module("fft", {"forward": True}, {"signal": self.multiply.output("product")},
                          p + "fft")
        self.agc = None
        spectrum = self.fft.output("signal")
        if enable_agc:  # one RMS tile per spectrum keeps the relative bin levels (:186-190)
            self.agc = Module("agc", {"tileSize": n}, {"signal": spectrum}, p + "agc")
            spectrum = self.agc.output("signal")
        self.amplitude = Module("amplitude", {}, {"signal": spectrum},
                                p + "amplitude", provider=provider)
        self.range = None
        if enable_scale:
            self.range = Module("range", {"min": range_min, "max": range_max},
                                {"signal": self.amplitude.output("signal")}, p + "range",
                                provider=provider)
            self.buffer = self.range.output("signal")
        else:
            self.buffer = self.amplitude.output("signal")

    @property
    def modules(self) -> List[Module]:
        ms = [self.cast, self.window, self.invert, self.reshape, self.multiply, self.fft]
        if self.agc is not None:
            ms.append(self.agc)
        ms.append(self.amplitude)
        if self.range is not None:
            ms.append(self.range)
        return ms


class _FilterPlanDesc(C.Structure):
    _fields_ = [("pad_size", C.c_uint64), ("convolution_size", C.c_uint64), ("resampler_size", C.c_uint64),
                ("resample", C.c_int32), ("resampled_sample_rate", C.c_float)]


_sig("jst_filter_plan", C.c_uint16, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_uint64,
     C.c_uint64, C.POINTER(_FilterPlanDesc), C.POINTER(C.c_uint64))


def filter_plan(sample_rate: float, bandwidth: float, center: Sequence[float], taps: int,
                heads: int, signal_size: int) -> dict:
    """CalculateCandidatePlan (src/domains/dsp/filter/block_impl.cc:40-168): convolution size, whether the block
    resamples (fold), per-head fold offsets (integers), pad size -- computed by the library (jst_filter_plan; host
    logic in csrc/modules/filter_modules.cc), this is only the ctypes call."""
    ctr = (C.c_float * max(len(center), 1))(*[float(c) for c in center])
    offs = (C.c_uint64 * max(heads, 1))()
    d = _FilterPlanDesc()
    _check(_lib.jst_filter_plan(float(sample_rate), float(bandwidth), ctr, len(center), int(taps), int(heads),
                                int(signal_size), C.byref(d), offs))
    return {"padSize": int(d.pad_size), "convolutionSize": int(d.convolution_size), "resample": bool(d.resample),
            "resamplerOffsets": [int(offs[h]) for h in range(heads)] if d.resample else [],
            "resamplerSize": int(d.resampler_size), "resampledSampleRate": float(d.resampled_sample_rate)}


class Filter:
    """The filter BLOCK: filter_taps -> pad -> fft -> multiply -> [fold] -> ifft -> normalize ->
    [phase_correction] -> unpad -> overlap_add, wired as src/domains/dsp/filter/block_impl.cc:350-582."""

    def __init__(self, signal: Tensor, sample_rate: float = 2.0e6, bandwidth: float = 1.0e6,
                 center: Sequence[float] = (0.0,), taps: int = 101, heads: int = 1,
                 name: str = "filter", provider: str = "generic"):
        """provider "fast": when every head is centred on 0 Hz the whole chain collapses into ONE
        direct-form polyphase FIR + decimate kernel (csrc/kernels/fir.hip) -- same ports, same stream
        continuity, results within the reference's own 1e-5-of-peak tolerance instead of bit-exact.
        Any other plan keeps the FFT overlap-add chain (``self.direct`` tells which one was built)."""
        import math
        axes = signal.axes
        rank = len(signal.shape)
        s_axis = axes["sample"] if axes["sample"] is not None else (0 if rank == 1 else None)
        if s_axis is None:
            raise JetstreamError(1, "[BLOCK_FILTER] Input validation plan is unavailable.")
        head_axis, sample_axis = s_axis, s_axis + 1
        signal_size = signal.shape[s_axis]
        self.plan = plan = filter_plan(sample_rate, bandwidth, center, taps, heads, signal_size)
        batch = axes["batch"]
        out_axes = {"sample": sample_axis, "channel": head_axis,
                    "batch": None if batch is None else (batch + 1 if batch >= head_axis else batch)}
        p = name + "."
        sr, bw = float(np.float32(sample_rate)), float(np.float32(bandwidth))
        ctr = [float(np.float32(c)) for c in list(center)[:heads]] + [0.0] * max(0, heads - len(center))
        self.filter_taps = Module("filter_taps", {"sampleRate": sr, "bandwidth": bw, "center": ctr,
                                                  "taps": taps}, {}, p + "filter_taps")
        filt = self.filter_taps.output("coeffs").set_axes(sample=1, channel=0)
        self.cast_signal = Module("cast", {"outputType": "CF32"}, {"buffer": signal}, p + "cast_signal")
        ratio = int(sr / bw) if plan["resample"] else 1
        self.direct = (provider == "fast" and all(c == 0.0 for c in ctr) and s_axis == rank - 1
                       and rank <= 2 and (rank == 1 or batch == 0))
        self.fir = self.fir_taps = None
        if self.direct:
            try:  # the kernel's own plan check decides (taps vs. row length, decimation, LDS tile)
                self.fir_taps = Module("fir_taps", {"decimation": ratio}, {"coeffs": filt}, p + "fir_taps",
                                       provider="fast")
                self.fir = Module("fir_decimate", {},
                                  {"signal": self.cast_signal.output("buffer"),
                                   "table": self.fir_taps.output("table")},
                                  p + "fir_decimate", provider="fast")
            except JetstreamError:
                self.direct, self.fir, self.fir_taps = False, None, None  # keep the FFT overlap-add chain
        if self.direct:
            self.buffer = self.fir.output("buffer").set_axes(**out_axes)
            if plan["resample"]:
                self.buffer.set_attribute("sampleRate", plan["resampledSampleRate"])
            return
        self.expand_signal = Module("expand_dims", {"axis": head_axis},
                                    {"buffer": self.cast_signal.output("buffer")}, p + "expand_signal")
        sig_in = self.expand_signal.output("buffer").set_axes(**out_axes)
        self.pad_signal = Module("pad", {"size": taps - 1, "axis": sample_axis}, {"unpadded": sig_in},
                                 p + "padSignal")
        self.pad_filter = Module("pad", {"size": signal_size - 1, "axis": 1}, {"unpadded": filt},
                                 p + "padFilter")
        self.fft_signal = Module("fft", {"forward": True},
                                 {"signal": self.pad_signal.output("padded")}, p + "fftSignal")
        self.fft_filter = Module("fft", {"forward": True},
                                 {"signal": self.pad_filter.output("padded")}, p + "fftFilter")
        spec_shape = [1] * (rank + 1)
        spec_shape[head_axis] = heads
        spec_shape[sample_axis] = plan["convolutionSize"]
        self.reshape_filter = Module("reshape", {"shape": spec_shape},
                                     {"buffer": self.fft_filter.output("signal")}, p + "reshape_filter")
        aligned = self.reshape_filter.output("buffer").set_axes(sample=sample_axis, channel=head_axis)
        self.multiply = Module("multiply", {}, {"a": self.fft_signal.output("signal"), "b": aligned},
                               p + "multiply")
        product = self.multiply.output("product").set_axes(**out_axes)
        ifft_in = product
        self.fold = None
        if plan["resample"]:
            product.set_attribute("channelOffsets", [int(o) for o in plan["resamplerOffsets"]])
            self.fold = Module("fold", {"offset": 0, "size": plan["resamplerSize"]},
                               {"buffer": product}, p + "fold")
            ifft_in = self.fold.output("buffer")
        self.ifft = Module("fft", {"forward": False}, {"signal": ifft_in}, p + "ifft")
        ifft_out = self.ifft.output("signal")
        n_ifft = ifft_out.shape[sample_axis]
        self.normalize = Module("multiply_constant", {"constant": float(np.float32(1.0) / np.float32(n_ifft))},
                                {"factor": ifft_out}, p + "normalize")
        normalized = self.normalize.output("product")
        self.phase_correction = None
        if plan["resample"] and any(o != 0 for o in plan["resamplerOffsets"]):
            inc = [math.remainder(2.0 * math.pi * float(o) * float(signal_size) /
                                  float(plan["convolutionSize"]), 2.0 * math.pi)
                   for o in plan["resamplerOffsets"]]
            normalized.set_attribute("channelPhaseIncrements", inc)
            self.phase_correction = Module("phase_correction", {"phaseIncrement": 0.0},
                                           {"signal": normalized}, p + "phase_correction")
            normalized = self.phase_correction.output("signal")
        self.unpad = self.overlap = None
        if plan["padSize"] == 0:
            self.buffer = normalized
        else:
            self.unpad = Module("unpad", {"size": plan["padSize"], "axis": sample_axis},
                                {"padded": normalized}, p + "unpad")
            self.overlap = Module("overlap_add", {}, {"buffer": self.unpad.output("unpadded"),
                                                      "overlap": self.unpad.output("pad")}, p + "overlap")
            self.buffer = self.overlap.output("buffer")
        self.buffer.set_axes(**out_axes)
        if plan["resample"]:
            self.buffer.set_attribute("sampleRate", plan["resampledSampleRate"])

    @property
    def modules(self) -> List[Module]:
        if self.direct:
            return [self.filter_taps, self.fir_taps, self.cast_signal, self.fir]
        ms = [self.filter_taps, self.cast_signal, self.expand_signal, self.pad_signal, self.pad_filter,
              self.fft_signal, self.fft_filter, self.reshape_filter, self.multiply, self.fold, self.ifft,
              self.normalize, self.phase_correction, self.unpad, self.overlap]
        return [m for m in ms if m is not None]


class FilterEngine:
    """The filter_engine BLOCK (src/domains/dsp/filter_engine/block_impl.cc:213-673): FFT overlap-add convolution of
    `signal` with an EXTERNAL coefficient tensor `filt` -- [T] with sampleAxis 0 (one head: no head axis is added, the
    fold offset and the phase increment are plain config values) or [C, T] with channelAxis 0 / sampleAxis 1 (C heads,
    per-head fold offsets / phase increments as tensor attributes).  Resampling is planned from the filter tensor's
    `sampleRate` / `bandwidth` / `center` attributes (what filter_taps publishes) and bypassed when one is missing
    (CalculateResampleHeuristics, :43-176; the integers come from the library's jst_filter_plan)."""

    def __init__(self, signal: Tensor, filt: Tensor, name: str = "filter_engine"):
        import math
        axes, rank = signal.axes, len(signal.shape)
        s_axis = axes["sample"] if axes["sample"] is not None else (0 if rank == 1 else None)
        faxes, frank = filt.axes, len(filt.shape)
        f_sample = faxes["sample"] if faxes["sample"] is not None else (0 if frank == 1 else None)
        if s_axis is None:
            raise JetstreamError(1, "[BLOCK_FILTER_ENGINE] Signal axis metadata is invalid.")
        if frank not in (1, 2):
            raise JetstreamError(1, "[BLOCK_FILTER_ENGINE] Filter input must be rank 1 or 2.")
        if (frank == 1 and (f_sample != 0 or faxes["batch"] is not None or faxes["channel"] is not None)) or \
           (frank == 2 and (f_sample != 1 or faxes["batch"] is not None or faxes["channel"] != 0)):
            raise JetstreamError(1, "[BLOCK_FILTER_ENGINE] Filter coefficients must be [T] with sampleAxis=0 or [C,T] "
                                    "with channelAxis=0 and sampleAxis=1.")
        multi = faxes["channel"] is not None
        if multi and axes["channel"] is not None:
            raise JetstreamError(1, "[BLOCK_FILTER_ENGINE] Cannot add filter channels to a signal that already "
                                    "carries channelAxis.")
        signal_size, filter_size = signal.shape[s_axis], filt.shape[f_sample]
        heads = filt.shape[0] if multi else 1

        def attr(key):
            try:
                return filt.attribute(key)
            except JetstreamError:
                return None
        sr, bw, ctr = attr("sampleRate"), attr("bandwidth"), attr("center")
        if ctr is not None:
            # a SCALAR F32 attribute expands across the filter channels (filter_engine/block_impl.cc:333-341); a vector -- of
            # length 1 included -- must match the channel extent (:342-345)
            if not isinstance(ctr, (list, tuple)):
                ctr = [ctr] * heads
            else:
                ctr = list(ctr)
                if len(ctr) != heads:
                    raise JetstreamError(1, "[BLOCK_FILTER_ENGINE] Filter center metadata must match the filter channel extent.")
        if sr is None or bw is None or ctr is None:   # bypass: plain convolution
            plan = {"padSize": filter_size - 1, "convolutionSize": signal_size + filter_size - 1, "resample": False,
                    "resamplerOffsets": [], "resamplerSize": 0, "resampledSampleRate": 0.0}
        else:
            plan = filter_plan(sr, bw, ctr, filter_size, heads, signal_size)
        self.plan = plan
        batch = axes["batch"]
        out_axes = dict(axes)
        out_axes["sample"] = s_axis
        if multi:
            out_axes = {"sample": s_axis + 1, "channel": s_axis,
                        "batch": None if batch is None else (batch + 1 if batch >= s_axis else batch)}
        sample_axis = s_axis + 1 if multi else s_axis
        p = name + "."
        self.cast_signal = Module("cast", {"outputType": "CF32"}, {"buffer": signal}, p + "cast_signal")
        self.cast_filter = Module("cast", {"outputType": "CF32"}, {"buffer": filt}, p + "cast_filter")
        sig_in = self.cast_signal.output("buffer")
        self.expand_signal = None
        if multi:
            self.expand_signal = Module("expand_dims", {"axis": s_axis}, {"buffer": sig_in}, p + "expand_signal")
            sig_in = self.expand_signal.output("buffer").set_axes(**out_axes)
        self.pad_signal = Module("pad", {"size": filter_size - 1, "axis": sample_axis}, {"unpadded": sig_in},
                                 p + "padSignal")
        self.pad_filter = Module("pad", {"size": signal_size - 1, "axis": f_sample},
                                 {"unpadded": self.cast_filter.output("buffer")}, p + "padFilter")
        self.fft_signal = Module("fft", {"forward": True}, {"signal": self.pad_signal.output("padded")}, p + "fftSignal")
        self.fft_filter = Module("fft", {"forward": True}, {"signal": self.pad_filter.output("padded")}, p + "fftFilter")
        fspec = self.fft_filter.output("signal")
        want = [1] * len(sig_in.shape)
        if multi:
            want[s_axis] = heads
        want[sample_axis] = plan["convolutionSize"]
        self.reshape_filter = None
        if list(fspec.shape) != want:
            self.reshape_filter = Module("reshape", {"shape": want}, {"buffer": fspec}, p + "reshape_filter")
            fspec = self.reshape_filter.output("buffer")
        fspec.set_axes(sample=sample_axis, channel=s_axis if multi else None)
        self.multiply = Module("multiply", {}, {"a": self.fft_signal.output("signal"), "b": fspec}, p + "multiply")
        product = self.multiply.output("product").set_axes(**out_axes)
        offsets = [int(o) for o in plan["resamplerOffsets"]]
        ifft_in, self.fold = product, None
        if plan["resample"]:
            if multi:
                product.set_attribute("channelOffsets", offsets)
            self.fold = Module("fold", {"offset": 0 if multi else offsets[0], "size": plan["resamplerSize"]},
                               {"buffer": product}, p + "fold")
            ifft_in = self.fold.output("buffer")
        self.ifft = Module("fft", {"forward": False}, {"signal": ifft_in}, p + "ifft")
        ifft_out = self.ifft.output("signal")
        self.normalize = Module("multiply_constant",
                                {"constant": float(np.float32(1.0) / np.float32(ifft_out.shape[sample_axis]))},
                                {"factor": ifft_out}, p + "normalize")
        normalized = self.normalize.output("product")
        self.phase_correction = None
        if plan["resample"] and any(o != 0 for o in offsets):
            inc = [math.remainder(2.0 * math.pi * float(o) * float(signal_size) / float(plan["convolutionSize"]),
                                  2.0 * math.pi) for o in offsets]
            if multi:
                normalized.set_attribute("channelPhaseIncrements", inc)
            self.phase_correction = Module("phase_correction", {"phaseIncrement": 0.0 if multi else inc[0]},
                                           {"signal": normalized}, p + "phase_correction")
            normalized = self.phase_correction.output("signal")
        self.unpad = self.overlap = None
        if plan["padSize"] == 0:
            self.buffer = normalized
        else:
            self.unpad = Module("unpad", {"size": plan["padSize"], "axis": sample_axis}, {"padded": normalized},
                                p + "unpad")
            self.overlap = Module("overlap_add", {}, {"buffer": self.unpad.output("unpadded"),
                                                      "overlap": self.unpad.output("pad")}, p + "overlap")
            self.buffer = self.overlap.output("buffer")
        self.buffer.set_axes(**out_axes)
        if plan["resample"]:
            self.buffer.set_attribute("sampleRate", plan["resampledSampleRate"])

    @property
    def modules(self) -> List[Module]:
        ms = [self.cast_signal, self.cast_filter, self.expand_signal, self.pad_signal, self.pad_filter,
              self.fft_signal, self.fft_filter, self.reshape_filter, self.multiply, self.fold, self.ifft,
              self.normalize, self.phase_correction, self.unpad, self.overlap]
        return [m for m in ms if m is not None]


class Decimator:
    """The decimator BLOCK: reshape [.., S/r, r] -> arithmetic(add, ratio axis) -> squeeze_dims ->
    duplicate (src/domains/dsp/decimator/block_impl.cc:140-207): integrate-and-dump, no divide."""

    def __init__(self, buffer: Tensor, ratio: int = 4, name: str = "decimator"):
        axes = buffer.axes
        shape = list(buffer.shape)
        s_axis = axes["sample"] if axes["sample"] is not None else (0 if len(shape) == 1 else None)
        if s_axis is None or ratio == 0 or shape[s_axis] % ratio != 0:
            raise JetstreamError(1, "[BLOCK_DECIMATOR] Input validation plan is unavailable.")
        child = s_axis + 1
        new_shape = shape[:s_axis] + [shape[s_axis] // ratio, ratio] + shape[s_axis + 1:]
        bump = lambda a: None if a is None else (a + 1 if a > s_axis else a)
        reshaped_axes = {k: bump(v) for k, v in axes.items()}
        reshaped_axes["sample"] = s_axis
        p = name + "."
        self.reshape = Module("reshape", {"shape": new_shape}, {"buffer": buffer}, p + "reshape")
        reshaped = self.reshape.output("buffer").set_axes(**reshaped_axes)
        self.arithmetic = Module("arithmetic", {"operation": "add", "axis": child},
                                 {"buffer": reshaped}, p + "arithmetic")
        self.squeeze = Module("squeeze_dims", {"axis": child},
                              {"buffer": self.arithmetic.output("buffer")}, p + "squeeze_dims")
        squeezed = self.squeeze.output("buffer").set_axes(sample=s_axis, batch=axes["batch"],
                                                          channel=axes["channel"])
        self.duplicate = Module("duplicate", {}, {"buffer": squeezed}, p + "duplicate")
        self.buffer = self.duplicate.output("buffer").set_axes(sample=s_axis, batch=axes["batch"],
                                                               channel=axes["channel"])

    @property
    def modules(self) -> List[Module]:
        return [self.reshape, self.arithmetic, self.squeeze, self.duplicate]
