"""GPU: the collective of the path EXECUTES.  A one-GPU box cannot host two ranks, so until now csrc/jst/comm.cc's RCCL
branch had never run: `world == 1` returned before RCCL was touched.  ncclCommInitRank(nranks = 1) is legal, and
jst_comm_init(0, 1, id) with an id now opens a REAL one-rank RCCL communicator: the dlopen, the symbol table, the
dtype / op enum slice (kNcclUint32 / kNcclFloat32, kNcclSum / kNcclMax), the in-place call on the module's own HBM, the
caller's stream and launch_divide_f32 all run on hardware here exactly as they do at world > 1 (the U32[H, N]
hit counts of the exact multi-GPU Spectrogram, the F32 trace of config 5's averaged spectrum)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(js):
    if not js.comm_available():
        pytest.skip("librccl.so is not loadable on this box")
    c = js.Comm(0, 1, js.comm_unique_id())
    assert c.uses_rccl and c.world == 1 and c.rank == 0
    return c


def test_world_one_without_an_id_stays_rccl_free(js):
    c = js.Comm(0, 1, None)
    assert not c.uses_rccl
    t = js.Tensor.from_numpy(np.arange(16, dtype=np.float32))
    c.all_reduce(t, "sum")
    assert c.calls == 1 and np.array_equal(t.numpy(), np.arange(16, dtype=np.float32))


def test_u32_hit_counts_sum_is_exact(js, comm):
    rng = np.random.default_rng(11)
    counts = rng.integers(0, 2**32, (256, 4096), dtype=np.uint64).astype(np.uint32)   # 4 MiB: config 2's [H, N]
    t = js.Tensor.from_numpy(counts)
    before = comm.calls
    comm.all_reduce(t, "sum")
    comm.all_reduce(t, "max")
    assert comm.calls == before + 2
    assert np.array_equal(t.numpy(), counts)     # one rank: the sum (and the max) of one contribution, bit for bit


def test_f32_trace_sum_and_average(js, comm):
    rng = np.random.default_rng(12)
    trace = rng.standard_normal(65536).astype(np.float32)       # config 5's averaged spectrum: F32[N = 65536]
    t = js.Tensor.from_numpy(trace)
    comm.all_reduce(t, "sum")
    assert np.array_equal(t.numpy().view(np.uint32), trace.view(np.uint32))
    comm.all_reduce(t, "sum", average=True)                     # sum, then launch_divide_f32 by the world size (1.0f)
    assert np.array_equal(t.numpy().view(np.uint32), trace.view(np.uint32))
    with pytest.raises(js.JetstreamError):
        comm.all_reduce(js.Tensor.from_numpy(np.zeros(8, np.uint32)), "sum", average=True)   # the average is F32-only


def test_collective_is_ordered_on_the_runtime_stream(js, comm, oracle):
    """The counts a `spectrogram{merge=counts}` module writes are reduced on the PRODUCER's stream with no host
    synchronisation in between, then applied by `spectrogram_merge`: the result must be the plain spectrogram's."""
    rng = np.random.default_rng(13)
    b, n, h = 64, 1024, 128
    xs = [rng.uniform(-0.1, 1.1, (b, n)).astype(np.float32) for _ in range(3)]
    sig = js.Tensor.from_numpy(xs[0], sample=1, batch=0)
    counts_mod = js.Module("spectrogram", {"height": h, "merge": "counts"}, {"signal": sig}, "counts")
    counts = counts_mod.output("counts")
    merge = js.Module("spectrogram_merge", {"batches": b}, {"counts": counts}, "merge")
    rt_a = js.Runtime([counts_mod])
    rt_b = js.Runtime([merge])
    ref_bins = np.zeros(n * h, dtype=np.float32)
    for x in xs:
        sig.copy_from(x)
        rt_a.compute(1, sync=False)
        comm.all_reduce(counts, "sum", stream=rt_a.stream)      # behind the counts kernel, same stream, no sync
        rt_a.synchronize()
        rt_b.compute(1)
        oracle.spectrogram(ref_bins, x, h)
    got = merge.state("frequencyBins").numpy().reshape(-1)
    assert np.array_equal(got.view(np.uint32), ref_bins.view(np.uint32))
