"""jst_tensor_rebind: a module's output storage moved onto memory the HOST FRAMEWORK allocated -- what the reference-side
DeviceType::HIP modules of integration/device_hip/ do, because the reference's Impl::create() has already allocated (and
published) the output tensor when the device binding runs (fft/module_impl.cc:80-83).  The kernel must then write into that
buffer, every view of the storage following, per module and inside a fused, graph-captured chain."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def test_fft_writes_into_the_callers_buffer(js, oracle):
    import torch
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((6, 4096)) + 1j * rng.standard_normal((6, 4096))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    fft = js.Module("fft", {"forward": True}, {"signal": src}, "fft")
    out = fft.output("signal")
    mine = torch.full((6, 4096), 7.0 + 0j, dtype=torch.complex64, device="cuda")   # "the framework's" allocation
    view = out.clone()                                   # a consumer's view taken BEFORE the rebind follows too
    out.rebind(mine.data_ptr(), mine.numel() * 8)
    rt = js.Runtime([fft], graph=True)
    rt.compute(3)
    torch.cuda.synchronize()
    want = oracle.fft_c2c(x, forward=True)
    assert_bit_equal(mine.cpu().numpy(), want, "the caller's buffer holds the transform")
    assert_bit_equal(view.numpy(), want, "a view taken before the rebind reads the same storage")
    rt.destroy()


def test_rebind_refuses_what_it_cannot_honour(js):
    import torch
    src = js.Tensor.from_numpy(np.zeros((2, 64), np.complex64), sample=1, batch=0)
    fft = js.Module("fft", {}, {"signal": src}, "fft")
    small = torch.zeros(8, dtype=torch.complex64, device="cuda")
    with pytest.raises(js.JetstreamError):
        fft.output("signal").rebind(small.data_ptr(), small.numel() * 8)      # smaller than the storage it replaces
    ring = js.Module("ring_source", {"batches": 2, "samples": 64, "slots": 3}, {}, "ring")
    big = torch.zeros(4096, dtype=torch.complex64, device="cuda")
    with pytest.raises(js.JetstreamError):
        ring.output("buffer").rebind(big.data_ptr(), big.numel() * 8)          # ring storage stays where it is


def test_rebind_refuses_planned_storage_and_foreign_memory(js):
    """a runtime's captured graphs, transform plans and ring selections hold raw addresses from Runtime::create on --
    the storage cannot move until the runtime is gone; and a HIP tensor cannot be moved onto memory the device cannot address."""
    import torch
    src = js.Tensor.from_numpy(np.ones((2, 64), np.complex64), sample=1, batch=0)
    fft = js.Module("fft", {}, {"signal": src}, "fft")
    mine = torch.zeros((2, 64), dtype=torch.complex64, device="cuda")
    host = np.zeros((2, 64), np.complex64)                       # pageable host memory: unknown to the HIP runtime
    with pytest.raises(js.JetstreamError, match="cannot address"):
        fft.output("signal").rebind(host.ctypes.data, host.nbytes)
    rt = js.Runtime([fft], graph=True)
    with pytest.raises(js.JetstreamError, match="planned over"):
        fft.output("signal").rebind(mine.data_ptr(), mine.numel() * 8)
    rt.destroy()
    fft.output("signal").rebind(mine.data_ptr(), mine.numel() * 8)   # the runtime is gone: the storage may move again
    rt = js.Runtime([fft])
    rt.compute(1)
    torch.cuda.synchronize()
    assert abs(mine.cpu().numpy()[0, 0] - 64.0) < 1e-3
    rt.destroy()


def test_view_keeps_the_storage_and_copy_moves_the_bytes(js):
    """jst_tensor_view: a consumer's geometry on a producer's storage (what a reference Tensor copy is after slice / permute);
    jst_tensor_copy: Tensor::copyFrom(source, stream)."""
    x = (np.arange(6 * 8, dtype=np.float32)).reshape(6, 8)
    t = js.Tensor.from_numpy(x)
    v = t.view((3, 4), (16, 2), 1)                   # rows 0, 2, 4; columns 1, 3, 5, 7
    assert tuple(v.shape) == (3, 4) and tuple(v.stride) == (16, 2) and v.offset == 1 and v.data_ptr == t.data_ptr
    m = js.Module("duplicate", {}, {"buffer": v}, "dense")     # a module that walks the view's strides
    rt = js.Runtime([m])
    rt.compute(1)
    assert np.array_equal(m.output("buffer").numpy(), x[0:6:2, 1:8:2])
    rt.destroy()
    with pytest.raises(js.JetstreamError, match="exceeds"):
        t.view((7, 8))
    dst = js.Tensor.create("hip", "F32", (6, 8))
    dst.copy_from_tensor(t)
    assert np.array_equal(dst.numpy(), x)
    with pytest.raises(js.JetstreamError):
        js.Tensor.create("hip", "F32", (5, 8)).copy_from_tensor(t)


def test_debug_switches_are_a_table_not_the_environment(js, monkeypatch):
    """jst_debug_set: the switch of a known name changes, an unknown name is an error; the environment variable of the same name,
    set AFTER the library asked once, changes nothing (nothing on a launch path calls getenv)."""
    js.debug_set("JST_FFT_KERNEL", "pipe")
    js.debug_set("JST_FFT_KERNEL", None)
    with pytest.raises(js.JetstreamError, match="unknown switch"):
        js.debug_set("JST_NO_SUCH_SWITCH", "1")
    monkeypatch.setenv("JST_FM_SERIAL", "1")              # too late: the table was seeded when the library first asked
    rng = np.random.default_rng(1)
    x = ((rng.standard_normal((2, 2311)) + 1j * rng.standard_normal((2, 2311))) * 0.4).astype(np.complex64)
    outs = []
    for forced in (None, "1"):
        js.debug_set("JST_FM_SERIAL", forced)
        t = js.Tensor.from_numpy(x, batch=0, sample=1)
        m = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 240e3}, {"signal": t})
        rt = js.Runtime([m])
        rt.compute(1)
        outs.append(m.output("signal").numpy().copy())
        rt.destroy()
    js.debug_set("JST_FM_SERIAL", None)
    assert_bit_equal(outs[0], outs[1], "the two FM walks agree (and both ran without touching the environment)")
