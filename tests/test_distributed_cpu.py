"""world_size-2 gloo tests (CPU) of the multi-GPU glue: batch sharding covers every batch once,
the averaged-spectrum all-reduce equals the single-process average, merged spectrogram hit
counts reproduce the single-process persistence display bit for bit, and the timing rule is
max-over-ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cyberether_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    rng = np.random.default_rng(5)
    total, n, h = 13, 256, 32
    x = rng.uniform(-0.1, 1.1, (total, n)).astype(np.float32)      # every rank sees the same stream
    start, count = D.shard_batches(total, rank, world)
    mine = x[start:start + count]

    # 1. averaged spectrum: per-rank lineplot trace, then ONE all-reduce
    avg = np.zeros(n, np.float32)
    oracle.lineplot(avg, mine, averaging=1)
    merged = D.allreduce_average(torch.from_numpy(avg.copy())).numpy()

    # 2. spectrogram: integer hit counts merge exactly
    counts = np.zeros((h, n), np.int64)
    f = mine * np.float32(h)
    hit = (f >= 1) & (f < h)
    idx = f.astype(np.int64)
    for b in range(count):
        cols = np.flatnonzero(hit[b])
        np.add.at(counts, (idx[b, cols], cols), 1)
    all_counts = D.merge_hit_counts(torch.from_numpy(counts.copy())).numpy()

    slow = D.max_over_ranks(1.0 + rank)
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), merged=merged, counts=all_counts, slow=slow,
             start=start, count=count, local_avg=avg)
    dist.destroy_process_group()


def test_two_rank_sharding_and_reductions(tmp_path, oracle):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    # shards tile the batch axis exactly once
    assert int(r[0]["start"]) == 0 and int(r[0]["count"]) + int(r[1]["count"]) == 13
    assert int(r[1]["start"]) == int(r[0]["count"])
    # both ranks hold the same merged results
    assert np.array_equal(r[0]["merged"], r[1]["merged"]) and np.array_equal(r[0]["counts"], r[1]["counts"])
    assert float(r[0]["slow"]) == float(r[1]["slow"]) == 2.0
    np.testing.assert_allclose(r[0]["merged"], (r[0]["local_avg"] + r[1]["local_avg"]) / 2, rtol=0, atol=1e-7)
    # merged counts reproduce the single-process spectrogram bit for bit
    rng = np.random.default_rng(5)
    x = rng.uniform(-0.1, 1.1, (13, 256)).astype(np.float32)
    ref = np.zeros(256 * 32, np.float32)
    oracle.spectrogram(ref, x, 32)
    got = D.apply_hit_counts(np.zeros((32, 256), np.float32), r[0]["counts"], oracle.spectrogram_decay(13))
    assert np.array_equal(got.reshape(-1).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("total,world", [(1024, 8), (13, 2), (5, 8), (0, 3)])
def test_shard_batches_partition(total, world):
    spans = [D.shard_batches(total, r, world) for r in range(world)]
    assert sum(c for _, c in spans) == total
    pos = 0
    for start, count in spans:
        assert start == pos
        pos += count
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_single_process_paths_are_identity():
    t = torch.ones(4)
    assert D.allreduce_average(t) is t and D.max_over_ranks(0.5) == 0.5
