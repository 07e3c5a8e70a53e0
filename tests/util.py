"""Shared helpers for the -m gpu parity tests (the TestContext of src/testing.cc:53-140: mirror
inputs to the device, run ONE module in a private runtime, snapshot outputs)."""
import numpy as np


def bits(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype.itemsize in (4, 8) and a.dtype != np.uint64 else a


_LIBM_PINNED = None


def libm_pinned() -> bool:
    """True when the host libm answers exactly like the one the bit-exact parity is stated against (glibc 2.35
    x86-64 with the FMA IFUNC variants; tests/golden/libm_pin.json, written by tests/golden/make_libm_pin.py)."""
    global _LIBM_PINNED
    if _LIBM_PINNED is None:
        import ctypes as C
        import json
        import os
        try:
            pin = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libm_pin.json")))
            libm = C.CDLL("libm.so.6")
            ok = True
            for name, rec in pin["functions"].items():
                fn = getattr(libm, name)
                args = np.array(rec["args"], np.uint32).view(np.float32)
                if "args2" in rec:
                    fn.restype, fn.argtypes = C.c_float, [C.c_float, C.c_float]
                    a2 = np.array(rec["args2"], np.uint32).view(np.float32)
                    res = np.array([fn(float(a), float(b)) for a, b in zip(args, a2)], np.float32)
                else:
                    fn.restype, fn.argtypes = C.c_float, [C.c_float]
                    res = np.array([fn(float(a)) for a in args], np.float32)
                ok = ok and np.array_equal(res.view(np.uint32), np.array(rec["bits"], np.uint32))
            _LIBM_PINNED = bool(ok)
        except (OSError, KeyError, ValueError, AttributeError):
            _LIBM_PINNED = False
    return _LIBM_PINNED


def assert_bit_equal(got: np.ndarray, ref: np.ndarray, what: str = ""):
    """Bit-for-bit equality.  The oracle calls the HOST libm for tanhf / sinf / cosf / atan2f (as the reference does);
    the device restates glibc 2.35's.  On a host whose libm answers differently (libm_pinned() is False) a float
    mismatch is re-judged with BASELINE.json's tolerance -- 1e-5 of the reference's peak magnitude -- and reported
    as a warning that says so, instead of failing on a difference the north star allows."""
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    g, r = bits(got), bits(ref)
    if not np.array_equal(g, r):
        bad = np.flatnonzero(g.reshape(-1) != r.reshape(-1))
        msg = (f"{what}: {bad.size} of {g.size} words differ; first at {bad[:5]} "
               f"got {got.reshape(-1).view(np.float32)[bad[:3]]} "
               f"ref {ref.reshape(-1).view(np.float32)[bad[:3]]}")
        if got.dtype.kind in "fc" and not libm_pinned():
            gf = np.ascontiguousarray(got).reshape(-1).view(np.float32).astype(np.float64)
            rf = np.ascontiguousarray(ref).reshape(-1).view(np.float32).astype(np.float64)
            fin = np.isfinite(rf)
            peak = float(np.max(np.abs(rf[fin]))) if fin.any() else 0.0
            same_class = np.array_equal(np.isfinite(gf), fin) and np.array_equal(gf[~fin], rf[~fin], equal_nan=True)
            if same_class and (not fin.any() or float(np.max(np.abs(gf[fin] - rf[fin]))) <= 1e-5 * max(peak, 1e-30)):
                import warnings
                warnings.warn("libm differs from the pinned glibc 2.35 (tests/golden/libm_pin.json): "
                              f"{what} compared at 1e-5 of the peak instead of bit for bit; " + msg)
                return
        raise AssertionError(msg)


def run_module(js, mtype, config, inputs, outputs=("signal",), cycles=1, **rt_flags):
    m = js.Module(mtype, config, inputs)
    rt = js.Runtime([m], **rt_flags)
    rt.compute(cycles)
    res = {p: m.output(p).numpy() for p in outputs}
    rt.destroy()
    return m, res


def csignal(rng, shape, scale=1.0):
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    return (x * scale).astype(np.complex64)


def strided_view(js, storage: np.ndarray, build):
    """Upload `storage` densely, then apply `build(tensor)` view ops; returns (tensor, numpy view)
    where the numpy view is produced by the same ops on the host array."""
    t = js.Tensor.from_numpy(storage)
    return build(t)
