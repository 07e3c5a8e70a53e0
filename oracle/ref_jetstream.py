"""ctypes front-end for oracle/_ref/libref_jetstream.so: the REFERENCE's own libjetstream core and native-CPU
modules / blocks, compiled IN PLACE from /root/reference by oracle/ref_jetstream_build.sh (harness: ref_jetstream.cc).

TEST INFRASTRUCTURE ONLY.  Used by tests/ to pin oracle/jst_oracle.c stage by stage against the reference itself and
by tools/make_reference_vectors.py to freeze golden vectors (tests/golden/reference_modules.npz) for boxes without the
reference tree.  The product package never imports this.

    with RefModule("fm", {"mode": "wide", "sampleRate": 200e3}) as m:
        m.input("signal", x, sample=0)        # numpy array; a CPU tensor of the reference holds a copy
        m.start(); m.compute()
        y = m.output("signal")                # numpy copy, layout as the reference's tensor
        m.write("signal", x2); m.compute()    # next submission: state carries over
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_jetstream.so")
# the same reference objects + integration/mi355x_provider, linked against cyberether_amd/lib/libjetstream_hip.so
# (oracle/ref_jetstream_build.sh): select with use_hip_library() BEFORE the first call, one library per process
_PATH_HIP = os.path.join(_HERE, "_ref", "libref_jetstream_hip.so")
# the reference's core PATCHED with DeviceType::HIP (integration/device_hip/) + the HIP buffer backend, runtime and modules:
# device-resident, select with use_device_hip_library() BEFORE the first call
_PATH_DEVHIP = os.path.join(_HERE, "_ref", "libref_jetstream_devhip.so")

RESULT_SUCCESS = 0

_DTYPES = {1: np.float32, 2: np.float64, 3: np.int8, 4: np.int16, 5: np.int32, 6: np.int64, 7: np.uint8, 8: np.uint16,
           9: np.uint32, 10: np.uint64, 11: np.complex64, 12: np.complex128}
_NAMES = {np.dtype(np.float32): "F32", np.dtype(np.float64): "F64", np.dtype(np.complex64): "CF32",
          np.dtype(np.complex128): "CF64", np.dtype(np.int8): "I8", np.dtype(np.int16): "I16", np.dtype(np.int32): "I32",
          np.dtype(np.uint8): "U8", np.dtype(np.uint16): "U16", np.dtype(np.uint32): "U32", np.dtype(np.uint64): "U64"}
# complex integer formats travel as [..., 2] integer arrays
_CI = {"CI8": (np.int8, 13), "CI16": (np.int16, 14), "CU8": (np.uint8, 17), "CU16": (np.uint16, 18)}


class _Desc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offset", C.c_uint64), ("dtype", C.c_uint32), ("rank", C.c_uint32),
                ("shape", C.c_uint64 * 8), ("stride", C.c_uint64 * 8),
                ("sample_axis", C.c_int64), ("batch_axis", C.c_int64), ("channel_axis", C.c_int64),
                ("device", C.c_uint64), ("buffer_bytes", C.c_uint64)]

DEVICE_CPU, DEVICE_HIP = 1 << 1, 1 << 6   # include/jetstream/memory/types.hh:22-29 (+ integration/device_hip/core_hip_device.patch)


_lib = None


def build(force: bool = False) -> None:
    if os.path.exists("/root/reference/src/module.cc") and (force or not os.path.exists(_PATH)):
        subprocess.check_call([os.path.join(_HERE, "ref_jetstream_build.sh"), "-j", "16"], stdout=subprocess.DEVNULL)


def available() -> bool:
    build()
    return os.path.exists(_PATH)


def hip_library_available() -> bool:
    return os.path.exists(_PATH_HIP) or os.path.exists(_PATH_DEVHIP)


def use_hip_library() -> None:
    """The reference linked against the HIP library (provider "mi355x" registered): for tests/test_gpu_reference_drives_library.py.
    The device-resident build holds the provider modules too and is preferred: ONE reference build per process."""
    global _PATH
    want = _PATH_DEVHIP if os.path.exists(_PATH_DEVHIP) else _PATH_HIP
    assert _lib is None or _PATH == want, "another build of the reference is already loaded in this process"
    _PATH = want


def device_hip_library_available() -> bool:
    return os.path.exists(_PATH_DEVHIP)


def use_device_hip_library() -> None:
    """The reference with DeviceType::HIP (patched core + integration/device_hip): for tests/test_gpu_reference_device_hip.py."""
    global _PATH
    assert _lib is None or _PATH == _PATH_DEVHIP, "another build of the reference is already loaded in this process"
    _PATH = _PATH_DEVHIP


def hip_runtime_configure(hand_off=True, defer_cycles: int = 0) -> None:
    """integration/device_hip/runtime_native_hip_impl.cc: knobs of the HIP runtimes created from here on (hand_off: False / 0 =
    module by module, True / 1 = library segments as one jst_runtime (fusion; hipGraph replay for cycle-batched spans, direct launches for
    single cycles), 2 = direct launches always, 3 = hipGraph always)."""
    lib().jetstream_hip_runtime_configure(C.c_int(int(hand_off)), C.c_uint64(defer_cycles))


def hip_runtime_flush() -> None:
    assert lib().jetstream_hip_runtime_flush() == 0, "a deferred span failed"


def hip_runtime_units() -> str:
    """Units of every live handed-off HIP runtime ('|' between runtimes, '[batched]' behind a cycle-batched one)."""
    l = lib()
    l.jetstream_hip_runtime_units.restype = C.c_size_t
    buf = C.create_string_buffer(8192)
    l.jetstream_hip_runtime_units(buf, C.c_size_t(8192))
    return buf.value.decode()


def hip_directory(module: str, port: str) -> np.ndarray:
    """The library tensor behind `port` (or "state:<key>") of the reference's HIP module `module`, read back (copy)."""
    d = _Desc()
    assert lib().ref_hip_directory(module.encode(), port.encode(), C.byref(d)) == 0, f"no library tensor for {module}:{port}"
    return _view(d)


def registry_has(mtype: str, provider: str, device: str = "cpu") -> bool:
    if device != "cpu":
        return bool(lib().ref_registry_has_on(mtype.encode(), provider.encode(), device.encode()))
    return bool(lib().ref_registry_has(mtype.encode(), provider.encode()))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if _PATH not in (_PATH_HIP, _PATH_DEVHIP):
            build()
        _lib = C.CDLL(_PATH)
        assert _lib.ref_jst_desc_size() == C.sizeof(_Desc)
        for n in ("ref_mod_new", "ref_fg_new"):
            getattr(_lib, n).restype = C.c_void_p
        for n in ("ref_mod_output_attr", "ref_fg_tensor_attr_get"):
            getattr(_lib, n).restype = C.c_uint64
        _lib.ref_jst_log_level(int(os.environ.get("JST_REF_LOG", "1")))
    return _lib


def _cfg(config: Optional[Dict]) -> bytes:
    def enc(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, (list, tuple, np.ndarray)):
            return "[" + ", ".join(enc(e) for e in v) + "]"
        if isinstance(v, (float, np.floating)):
            return repr(float(v))
        return str(v)
    return "\n".join(f"{k}={enc(v)}" for k, v in (config or {}).items()).encode()


def _device_image(d: _Desc, nbytes: int) -> int:
    """Host copy of the first `nbytes` of a device tensor's buffer; returns its address (kept alive on the descriptor)."""
    host = np.empty(max(nbytes, 1), np.uint8)
    assert lib().ref_dev_read(C.c_void_p(d.data), host.ctypes.data_as(C.c_void_p), C.c_uint64(nbytes)) == 0, "device read failed"
    d._host = host
    return host.ctypes.data


def _view(d: _Desc) -> np.ndarray:
    """numpy view (no copy) of a reference tensor: element strides and offset as Tensor::stride() / offset().  A tensor on
    another device than the CPU is read back first (ref_dev_read): the result is then a view of a host COPY."""
    code = int(d.dtype)
    if int(d.device) not in (0, DEVICE_CPU) and d.data:
        h = _Desc()
        C.memmove(C.byref(h), C.byref(d), C.sizeof(_Desc))
        h.data = _device_image(d, int(d.buffer_bytes))
        h.device = DEVICE_CPU
        return _Kept(_view(h), d._host)
    rank = int(d.rank)
    shape = [int(d.shape[a]) for a in range(rank)]
    stride = [int(d.stride[a]) for a in range(rank)]
    ci = {c: t for t, c in _CI.values()}
    if code in _DTYPES:
        dt = np.dtype(_DTYPES[code])
    elif code in ci:  # complex integer samples: an integer array whose last axis holds (re, im)
        dt = np.dtype(ci[code])
        shape, stride = shape + [2], [2 * st for st in stride] + [1]
        d_off = 2 * int(d.offset)
        extent = 1 + sum((s - 1) * st for s, st in zip(shape, stride)) if all(shape) else 0
        if extent == 0:
            return np.empty(shape, dt)
        base = np.ctypeslib.as_array(C.cast(C.c_void_p(d.data + 0), C.POINTER(C.c_uint8)), ((d_off + extent) * dt.itemsize,))
        flat = base.view(dt)[d_off:]
        return np.lib.stride_tricks.as_strided(flat, shape=shape, strides=[st * dt.itemsize for st in stride])
    else:
        raise TypeError(f"dtype code {code}")
    extent = 1 + sum((s - 1) * st for s, st in zip(shape, stride)) if all(shape) else 0
    if extent == 0:
        return np.empty(shape, dt)
    base = np.ctypeslib.as_array(C.cast(C.c_void_p(d.data + 0), C.POINTER(C.c_uint8)), ((int(d.offset) + extent) * dt.itemsize,))
    flat = base.view(dt)[int(d.offset):]
    return np.lib.stride_tricks.as_strided(flat, shape=shape, strides=[st * dt.itemsize for st in stride])


def _upload(d: _Desc, x: np.ndarray, byte_offset: int = 0) -> None:
    """Dense host array -> the buffer of a device tensor (at byte_offset), through ref_dev_write."""
    x = np.ascontiguousarray(x)
    if x.nbytes == 0:
        return
    assert byte_offset + x.nbytes <= int(d.buffer_bytes), "upload exceeds the device buffer"
    assert lib().ref_dev_write(C.c_void_p(d.data + byte_offset), x.ctypes.data_as(C.c_void_p), C.c_uint64(x.nbytes)) == 0, "device write failed"


class _Kept(np.ndarray):
    """ndarray that keeps the host image it views alive."""
    def __new__(cls, view, keep):
        obj = view.view(cls)
        obj._keep = keep
        return obj

    def __array_finalize__(self, obj):
        self._keep = getattr(obj, "_keep", None)


def _attr_args(value):
    if isinstance(value, (list, tuple, np.ndarray)):
        arr = np.asarray(value, dtype=np.float64).reshape(-1)
        return arr, len(arr)
    return np.asarray([value], dtype=np.float64), 1


ATTR_INDEX, ATTR_F32, ATTR_VEC_F32, ATTR_VEC_U64, ATTR_VEC_F64, ATTR_F64 = range(6)


class RefModule:
    """One module of the reference behind Registry::BuildModule / Module::create / Runtime::compute."""

    def __init__(self, mtype: str, config: Optional[Dict] = None, provider: str = "generic", device: str = "cpu"):
        self._l = lib()
        self._h = C.c_void_p(self._l.ref_mod_new(mtype.encode(), _cfg(config)))
        if provider != "generic":
            self._l.ref_mod_set_provider(self._h, provider.encode())
        self._device = device
        if device != "cpu":   # the registry's device key: inputs are allocated there, the runtime is that device's
            assert self._l.ref_mod_set_device(self._h, device.encode()) == 0, f"unknown device {device}"
        self._in: Dict[str, np.ndarray] = {}
        self._in_desc: Dict[str, _Desc] = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._h:
            self._l.ref_mod_free(self._h)
            self._h = None

    def input(self, port: str, x: np.ndarray, sample=None, batch=None, channel=None, attrs: Optional[Dict] = None):
        x = np.asarray(x)
        shape = (C.c_uint64 * max(x.ndim, 1))(*x.shape)
        d = _Desc()
        assert self._l.ref_mod_input(self._h, port.encode(), _NAMES[x.dtype].encode(), x.ndim, shape, C.byref(d)) == 0
        if self._device == "cpu":
            v = _view(d)
            v[...] = x
        else:   # a device tensor: the host keeps a shadow, write() uploads it
            v = np.ascontiguousarray(x).copy()
            _upload(d, v)
        self._in[port] = v
        self._in_desc[port] = d
        for key, val in (("sampleAxis", sample), ("batchAxis", batch), ("channelAxis", channel)):
            if val is not None:
                self.input_attr(port, key, ATTR_INDEX, val)
        for key, (kind, val) in (attrs or {}).items():
            self.input_attr(port, key, kind, val)
        return v

    def input_ci(self, port: str, x: np.ndarray, dtype: str):
        """A complex-integer input ("CI8", "CI16", "CU8", "CU16"): x is an integer array whose last axis holds (re, im)."""
        x = np.ascontiguousarray(x, dtype=_CI[dtype][0])
        assert x.shape[-1] == 2
        shape = (C.c_uint64 * max(x.ndim - 1, 1))(*x.shape[:-1])
        d = _Desc()
        assert self._l.ref_mod_input(self._h, port.encode(), dtype.encode(), x.ndim - 1, shape, C.byref(d)) == 0
        if self._device == "cpu":
            v = _view(d)
            v[...] = x
        else:
            v = x.copy()
            _upload(d, v)
        self._in[port] = v
        self._in_desc[port] = d
        return v

    def input_attr(self, port: str, key: str, kind: int, value):
        arr, n = _attr_args(value)
        assert self._l.ref_mod_input_attr(self._h, port.encode(), key.encode(), kind,
                                          arr.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(n)) == 0

    def input_view(self, port: str, op: str, values: Sequence[int]):
        """permute / reshape / expand_dims / broadcast of an input before start() (Tensor::permute etc.)."""
        code = {"permute": 0, "reshape": 1, "expand_dims": 2, "broadcast": 3, "slice": 4}[op]
        v = (C.c_uint64 * max(len(values), 1))(*values)
        d = _Desc()
        assert self._l.ref_mod_input_view(self._h, port.encode(), code, v, C.c_uint64(len(values)), C.byref(d)) == 0
        self._in[port] = _view(d)
        return self._in[port]

    def write(self, port: str, x: np.ndarray):
        self._in[port][...] = x
        if self._device != "cpu":
            _upload(self._in_desc[port], self._in[port])

    def start(self) -> int:
        return int(self._l.ref_mod_start(self._h))

    def compute(self) -> int:
        return int(self._l.ref_mod_compute(self._h))

    def output_view(self, port: str) -> np.ndarray:
        d = _Desc()
        assert self._l.ref_mod_output(self._h, port.encode(), C.byref(d)) == 0, f"no output {port}"
        return _view(d)

    def output(self, port: str) -> np.ndarray:
        return np.array(self.output_view(port))

    def output_axes(self, port: str):
        d = _Desc()
        assert self._l.ref_mod_output(self._h, port.encode(), C.byref(d)) == 0
        f = lambda a: None if a < 0 else int(a)
        return {"sample": f(d.sample_axis), "batch": f(d.batch_axis), "channel": f(d.channel_axis)}

    def output_attr(self, port: str, key: str, kind: int):
        buf = np.zeros(64, np.float64)
        n = int(self._l.ref_mod_output_attr(self._h, port.encode(), key.encode(), kind,
                                            buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(64)))
        if n == 0:
            return None
        return buf[:n].tolist() if kind in (ATTR_VEC_F32, ATTR_VEC_U64) else buf[0]

    def run(self) -> int:
        r = self.start()
        return r if r != RESULT_SUCCESS else self.compute()

    def state(self, name: str) -> np.ndarray:
        """A visualization module's state tensor (spectrogram / waterfall "frequencyBins", lineplot "signalPoints"), read the
        way the reference's own tests read it (a derived accessor for the protected member): numpy copy."""
        d = _Desc()
        assert self._l.ref_mod_state(self._h, name.encode(), C.byref(d)) == 0, f"no state {name}"
        return np.array(_view(d))

    def state_scalar(self, name: str) -> int:
        self._l.ref_mod_state_scalar.restype = C.c_int64
        return int(self._l.ref_mod_state_scalar(self._h, name.encode()))


class RefFlowgraph:
    """A Flowgraph of the reference's BLOCKS (Flowgraph::blockCreate / compute, as tests/support/flowgraph_fixture.hh)."""

    def __init__(self):
        self._l = lib()
        h = self._l.ref_fg_new()
        assert h, "Flowgraph::create failed"
        self._h = C.c_void_p(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._h:
            self._l.ref_fg_free(self._h)
            self._h = None

    def source(self, name: str, x: np.ndarray, sample=None, batch=None, channel=None) -> np.ndarray:
        """oracle_source block holding a copy of x; returns the live view (write into it between computes)."""
        x = np.asarray(x)
        assert self.block(name, "oracle_source", {"shape": list(x.shape), "dataType": _NAMES[x.dtype]}) == 0
        v = self.tensor(name, "signal")
        v[...] = x
        for key, val in (("sampleAxis", sample), ("batchAxis", batch), ("channelAxis", channel)):
            if val is not None:
                self.set_attr(name, "signal", key, ATTR_INDEX, val)
        return v

    def block(self, name: str, btype: str, config: Optional[Dict] = None, inputs: Optional[Dict[str, str]] = None,
              provider: str = "generic", device: str = "cpu") -> int:
        ins = "\n".join(f"{port}={src}" for port, src in (inputs or {}).items()).encode()
        if device != "cpu":
            return int(self._l.ref_fg_block_on(self._h, name.encode(), btype.encode(), _cfg(config), ins, provider.encode(), device.encode()))
        if provider != "generic":
            return int(self._l.ref_fg_block_provider(self._h, name.encode(), btype.encode(), _cfg(config), ins, provider.encode()))
        return int(self._l.ref_fg_block(self._h, name.encode(), btype.encode(), _cfg(config), ins))

    def state(self, name: str) -> int:
        return int(self._l.ref_fg_block_state(self._h, name.encode()))

    def desc(self, block: str, port: str) -> _Desc:
        d = _Desc()
        assert self._l.ref_fg_tensor(self._h, block.encode(), port.encode(), C.byref(d)) == 0, f"no {block}:{port}"
        return d

    def ring_source(self, name: str, batches: int, samples: int, slots: int = 1, provider: str = "generic") -> int:
        """integration/device_hip/modules/ring_source.cc: `slots` batches resident in HBM, a cycle selects the next one."""
        return self.block(name, "ring_source", {"batches": batches, "samples": samples, "slots": slots}, provider=provider, device="hip")

    def ring_write(self, name: str, slot: int, x: np.ndarray) -> None:
        """Fill ring slot `slot` (the reference's tensor borrows slot 0: the slots follow one another in the allocation)."""
        d = self.desc(name, "buffer")
        x = np.ascontiguousarray(x)
        d.buffer_bytes = (slot + 1) * x.nbytes     # the borrowed tensor knows one slot; the allocation holds them all
        _upload(d, x, slot * x.nbytes)

    def tensor(self, block: str, port: str) -> np.ndarray:
        d = _Desc()
        assert self._l.ref_fg_tensor(self._h, block.encode(), port.encode(), C.byref(d)) == 0, f"no {block}:{port}"
        return _view(d)

    def axes(self, block: str, port: str):
        d = _Desc()
        assert self._l.ref_fg_tensor(self._h, block.encode(), port.encode(), C.byref(d)) == 0
        f = lambda a: None if a < 0 else int(a)
        return {"sample": f(d.sample_axis), "batch": f(d.batch_axis), "channel": f(d.channel_axis)}

    def set_attr(self, block: str, port: str, key: str, kind: int, value):
        arr, n = _attr_args(value)
        assert self._l.ref_fg_tensor_attr(self._h, block.encode(), port.encode(), key.encode(), kind,
                                          arr.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(n)) == 0

    def get_attr(self, block: str, port: str, key: str, kind: int):
        buf = np.zeros(64, np.float64)
        n = int(self._l.ref_fg_tensor_attr_get(self._h, block.encode(), port.encode(), key.encode(), kind,
                                               buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(64)))
        if n == 0:
            return None
        return buf[:n].tolist() if kind in (ATTR_VEC_F32, ATTR_VEC_U64) else buf[0]

    def compute_timed(self, epoch_s: float, epochs: int):
        """Flowgraph::compute() in a C loop, nanobench-style: computes per second of every epoch."""
        rates = (C.c_double * epochs)()
        rc = int(self._l.ref_fg_compute_timed(self._h, C.c_double(epoch_s), C.c_uint32(epochs), rates))
        assert rc == 0, f"Flowgraph::compute failed ({rc})"
        return [float(r) for r in rates]

    def view(self, block: str, port: str, op: str, values: Sequence[int]):
        code = {"permute": 0, "reshape": 1, "expand_dims": 2}[op]
        v = (C.c_uint64 * max(len(values), 1))(*values)
        assert self._l.ref_fg_tensor_view(self._h, block.encode(), port.encode(), code, v, C.c_uint64(len(values))) == 0

    def compute(self) -> int:
        return int(self._l.ref_fg_compute(self._h))

    def compute_n(self, n: int) -> int:
        """n x Flowgraph::compute() in a C loop."""
        return int(self._l.ref_fg_compute_n(self._h, C.c_uint64(n)))
