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
