"""HIP FFT module vs the oracle (bit-exact) -- sizes, directions, batches, layouts.
Mirrors the matrix of src/domains/dsp/fft/module_tests.cc (DC spike :52-93, roundtrip :95-147,
batched / strided / permuted / offset variants :225-271,395-443,445-536,596-700)."""
import numpy as np
import pytest

from util import assert_bit_equal, csignal, run_module

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["pipe", "slot"])
def fft_kernel_variant(request, switch):
    """Every test here runs against both FFT kernel variants (pipelined / slot), same bits."""
    switch("JST_FFT_KERNEL", request.param)
    yield


@pytest.mark.parametrize("m", range(0, 15))
@pytest.mark.parametrize("forward", [True, False])
def test_c2c_bit_exact_all_sizes(js, oracle, m, forward):
    n = 1 << m
    rng = np.random.default_rng(1000 + m)
    batch = 7 if n <= 2048 else 3
    x = csignal(rng, (batch, n))
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    _, out = run_module(js, "fft", {"forward": forward}, {"signal": src})
    assert_bit_equal(out["signal"], oracle.fft_c2c(x, forward), f"n={n} fwd={forward}")


def test_rank1_default_axis_and_dc_spike(js, oracle):
    x = np.ones(64, np.complex64)
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(x)})  # rank 1: sampleAxis 0
    y = out["signal"]
    assert abs(y[0].real - 64) < 1e-3 and np.max(np.abs(y[1:])) < 1e-3
    assert_bit_equal(y, oracle.fft_c2c(x))


def test_roundtrip_is_unnormalised(js):
    rng = np.random.default_rng(5)
    x = csignal(rng, (4, 1024))
    src = js.Tensor.from_numpy(x, sample=1, batch=0)
    f = js.Module("fft", {"forward": True}, {"signal": src}, "fwd")
    i = js.Module("fft", {"forward": False}, {"signal": f.output("signal")}, "inv")
    rt = js.Runtime([f, i])
    rt.compute()
    back = i.output("signal").numpy()
    assert np.max(np.abs(back - 1024 * x)) < 1e-2 * 1024 / 64
    assert f.output("signal").axes == {"sample": 1, "batch": 0, "channel": None}  # propagated


def test_large_batch_grid_stride(js, oracle):
    rng = np.random.default_rng(6)
    x = csignal(rng, (9000, 64))  # more transforms than the launch's workgroup cap covers at once
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    assert_bit_equal(out["signal"], oracle.fft_c2c(x))


def test_transform_along_leading_axis(js, oracle):
    # sampleAxis = 0 of a [N, B] tensor: element stride B along the transform
    rng = np.random.default_rng(7)
    x = csignal(rng, (256, 5))
    src = js.Tensor.from_numpy(x, sample=0, batch=1)
    _, out = run_module(js, "fft", {}, {"signal": src})
    ref = np.ascontiguousarray(oracle.fft_c2c(np.ascontiguousarray(x.T)).T)
    assert_bit_equal(out["signal"], ref)


def test_strided_offset_permuted_views(js, oracle):
    rng = np.random.default_rng(8)
    store = csignal(rng, (6, 4, 512))
    t = js.Tensor.from_numpy(store)
    t.slice(0, 1, 6, 2).slice(2, 128, 384, 1).permute((1, 0, 2)).set_axes(sample=2)
    host = np.ascontiguousarray(store[1:6:2, :, 128:384].transpose(1, 0, 2))
    assert t.shape == host.shape == (4, 3, 256) and t.offset == 4 * 512 + 128
    _, out = run_module(js, "fft", {}, {"signal": t})
    assert_bit_equal(out["signal"], oracle.fft_c2c(host))
    # every-other-sample view along the transform axis
    u = js.Tensor.from_numpy(store)
    u.slice(2, 0, 512, 2).set_axes(sample=2, batch=0, channel=1)
    _, out = run_module(js, "fft", {}, {"signal": u})
    assert_bit_equal(out["signal"], oracle.fft_c2c(np.ascontiguousarray(store[:, :, ::2])))


def test_rank4_outer_axes(js, oracle):
    rng = np.random.default_rng(9)
    x = csignal(rng, (2, 3, 4, 128))
    src = js.Tensor.from_numpy(x, sample=3, batch=0, channel=1)
    _, out = run_module(js, "fft", {}, {"signal": src})
    assert_bit_equal(out["signal"], oracle.fft_c2c(x))


def test_special_values_propagate_like_the_cpu(js, oracle):
    x = np.zeros((2, 16), np.complex64)
    x[0, 3] = np.inf
    x[1, 5] = complex(np.nan, 1.0)
    x[1, 0] = -0.0
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    ref = oracle.fft_c2c(x)
    got = out["signal"]
    assert np.array_equal(np.isnan(got.view(np.float32)), np.isnan(ref.view(np.float32)))
    fin = ~np.isnan(ref.view(np.float32))
    assert np.array_equal(got.view(np.uint32)[fin], ref.view(np.uint32)[fin])


@pytest.mark.parametrize("n", [3, 5, 6, 12, 15, 60, 100, 243, 625, 1000, 2000, 6000, 32768, 65536, 160000])
def test_general_lengths_bit_exact(js, oracle, n):
    """Lengths with factors 3 and 5 and lengths beyond LDS: pass-per-launch path (fft_global.hip)."""
    rng = np.random.default_rng(n)
    batch = 3 if n < 10000 else 2
    x = csignal(rng, (batch, n))
    for forward in (True, False):
        _, out = run_module(js, "fft", {"forward": forward},
                            {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
        assert_bit_equal(out["signal"], oracle.fft_c2c(x, forward), f"n={n} fwd={forward}")


def test_general_length_strided(js, oracle):
    rng = np.random.default_rng(77)
    store = csignal(rng, (4, 3, 300))
    t = js.Tensor.from_numpy(store)
    t.slice(2, 0, 300, 2).permute((1, 0, 2)).set_axes(sample=2)   # n = 150 = 2*3*5*5
    host = np.ascontiguousarray(store[:, :, ::2].transpose(1, 0, 2))
    _, out = run_module(js, "fft", {}, {"signal": t})
    assert_bit_equal(out["signal"], oracle.fft_c2c(host))
    lead = csignal(rng, (45, 4))                                  # transform along axis 0
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(lead, sample=0, batch=1)})
    assert_bit_equal(out["signal"], np.ascontiguousarray(oracle.fft_c2c(np.ascontiguousarray(lead.T)).T))


@pytest.mark.parametrize("n", [7, 11, 14, 49, 77, 121, 154, 1001, 2401, 8050,      # pass7 / pass11
                               13, 17, 23, 26, 46, 97, 169, 299, 1147, 2209, 8170,  # generic radix (passg)
                               53 * 4, 211 * 2, 4099, 8191, 8292, 10007])           # Bluestein
def test_every_pocketfft_plan_bit_exact(js, oracle, n):
    """Radix 7 / 11, the generic odd radix and Bluestein: whatever plan pocketfft_c picks for the
    length (pocketfft.hh:2472-2489) is the plan the device runs, bit for bit."""
    rng = np.random.default_rng(n)
    x = csignal(rng, (3, n))
    for forward in (True, False):
        _, out = run_module(js, "fft", {"forward": forward},
                            {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
        assert_bit_equal(out["signal"], oracle.fft_c2c(x, forward), f"n={n} fwd={forward}")


@pytest.mark.parametrize("n,path", [(8050, "tile"), (805, "tile"), (1016, "tile"), (2 * 61 * 61, "tile"),   # 23 | 23 | 127 | 61, 61
                                    (13 * 4096, "tile_pair"), (23 * 4096, "tile_pair"), (31 * 31 * 32, "tile_pair"),
                                    (2 * 127 * 127, "tile_pair"), (131 * 8, "passes")])
def test_generic_radix_runs_on_lds_tiles(js, oracle, n, path):
    """passg (pocketfft.hh:1314-1421) inside the tiled kernels (fft_tiled.hip: tile_pass_generic) for every prime
    13..127: one kernel up to 8192 points, columns + blocks kernels beyond -- the generic pass lands in either of the
    two -- and only a larger prime falls back to one launch per pass.  Bit for bit pocketfft in both directions."""
    assert js.fft_path(n) == path
    rng = np.random.default_rng(n)
    x = csignal(rng, (5, n))
    for forward in (True, False):
        _, out = run_module(js, "fft", {"forward": forward}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
        assert_bit_equal(out["signal"], oracle.fft_c2c(x, forward), f"n={n} fwd={forward}")


def test_generic_radix_tile_with_many_and_strided_transforms(js, oracle):
    """More transforms than one workgroup's lanes (several transforms per tile, ragged last tile) and a transform axis
    that is not the innermost one."""
    rng = np.random.default_rng(77)
    n = 13 * 16                                              # 208 points: up to 32 transforms per tile
    x = csignal(rng, (2500, n))
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    assert_bit_equal(out["signal"], oracle.fft_c2c(x))
    lead = csignal(rng, (17 * 6, 37))                        # transform along axis 0, n = 102 = 2*3*17
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(lead, sample=0, batch=1)})
    assert_bit_equal(out["signal"], np.ascontiguousarray(oracle.fft_c2c(np.ascontiguousarray(lead.T)).T))


def test_bluestein_strided_and_special_values(js, oracle):
    rng = np.random.default_rng(5)
    n = 422                                                # 2 * 211 -> Bluestein, n2 = 847 = 7*11*11 (odd)
    assert oracle.fft_bluestein_size(n) == 847
    lead = csignal(rng, (n, 5))                            # transform along axis 0 (strided both ways)
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(lead, sample=0, batch=1)})
    assert_bit_equal(out["signal"], np.ascontiguousarray(oracle.fft_c2c(np.ascontiguousarray(lead.T)).T))
    x = csignal(rng, (2, n))
    x[0, 0] = complex(np.inf, 0.0)                         # akf[0]*0 pads with NaN like the CPU path
    _, out = run_module(js, "fft", {}, {"signal": js.Tensor.from_numpy(x, sample=1, batch=0)})
    ref = oracle.fft_c2c(x)
    got = out["signal"]
    assert np.array_equal(np.isnan(got.view(np.float32)), np.isnan(ref.view(np.float32)))
    fin = ~np.isnan(ref.view(np.float32))
    assert np.array_equal(got.view(np.uint32)[fin], ref.view(np.uint32)[fin])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 25, 27, 32, 45, 64, 100, 125, 128, 243, 360,
                               1000, 1024, 4096, 6000, 8100, 65536,       # rfftp radices 2/3/4/5
                               191, 211, 257, 401, 4099, 8191,                 # pocketfft_r picks Bluestein
                               7, 11, 13, 14, 21, 22, 26, 35, 49, 77, 91, 98, 121, 143, 169, 182, 343, 1001,
                               2401, 8050, 30030, 44100])                      # generic radix (radfg / radbg)
def test_real_input_transforms_bit_exact(js, oracle, n):
    """F32 input (fft/module_impl_native_cpu.cc:142-167): r2r_fftpack forward and backward
    (FFTPACK halfcomplex) and r2c (complexOutput), all bit-identical to pocketfft's rfftp."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n)).astype(np.float32)
    t = lambda: js.Tensor.from_numpy(x, sample=1, batch=0)
    m, out = run_module(js, "fft", {"forward": True}, {"signal": t()})
    assert out["signal"].dtype == np.float32
    assert_bit_equal(out["signal"], oracle.fft_r2r(x, True), f"r2r forward n={n}")
    _, out = run_module(js, "fft", {"forward": False}, {"signal": t()})
    assert_bit_equal(out["signal"], oracle.fft_r2r(x, False), f"r2r backward n={n}")
    m, out = run_module(js, "fft", {"forward": True, "complexOutput": True}, {"signal": t()})
    assert out["signal"].dtype == np.complex64 and out["signal"].shape == (3, n // 2 + 1)
    assert_bit_equal(out["signal"], oracle.fft_r2c(x), f"r2c n={n}")
    # complexOutput is only honoured for forward transforms of real input (fft/module_impl.cc:33-38)
    m, out = run_module(js, "fft", {"forward": False, "complexOutput": True}, {"signal": t()})
    assert out["signal"].dtype == np.float32 and out["signal"].shape == (3, n)


def test_real_input_reference_kats_and_layouts(js, oracle):
    """fft/module_tests.cc:149-223 (FFTPACK real forward / backward KATs) + a strided leading axis."""
    _, out = run_module(js, "fft", {"forward": True}, {"signal": js.Tensor.from_numpy(np.array([1, 2, 3, 4], np.float32))})
    assert np.allclose(out["signal"], [10.0, -2.0, 2.0, -2.0], atol=1e-4)
    _, out = run_module(js, "fft", {"forward": False},
                        {"signal": js.Tensor.from_numpy(np.array([10, -2, 2, -2], np.float32))})
    assert np.allclose(out["signal"], [4.0, 8.0, 12.0, 16.0], atol=1e-4)       # unnormalised inverse
    rng = np.random.default_rng(1)
    lead = rng.standard_normal((60, 4)).astype(np.float32)                      # transform along axis 0
    _, out = run_module(js, "fft", {"forward": True, "complexOutput": True},
                        {"signal": js.Tensor.from_numpy(lead, sample=0, batch=1)})
    assert_bit_equal(out["signal"], np.ascontiguousarray(oracle.fft_r2c(np.ascontiguousarray(lead.T)).T))


def test_unsupported_cases_fail_loudly(js):
    with pytest.raises(js.JetstreamError, match="not implemented"):
        js.Module("fft", {}, {"signal": js.Tensor.from_numpy(np.zeros((2, 16), np.float64), sample=1, batch=0)})
