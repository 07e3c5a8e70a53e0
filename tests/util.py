"""Shared helpers for the -m gpu parity tests (the TestContext of src/testing.cc:53-140: mirror
inputs to the device, run ONE module in a private runtime, snapshot outputs)."""
import numpy as np


def bits(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype.itemsize in (4, 8) and a.dtype != np.uint64 else a


def assert_bit_equal(got: np.ndarray, ref: np.ndarray, what: str = ""):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    g, r = bits(got), bits(ref)
    if not np.array_equal(g, r):
        bad = np.flatnonzero(g.reshape(-1) != r.reshape(-1))
        raise AssertionError(f"{what}: {bad.size} of {g.size} words differ; first at {bad[:5]} "
                             f"got {got.reshape(-1).view(np.float32)[bad[:3]]} "
                             f"ref {ref.reshape(-1).view(np.float32)[bad[:3]]}")


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
