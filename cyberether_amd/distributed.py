"""Multi-GPU glue for the hot path: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by INDEPENDENT batches / streams: every rank runs the whole
Window -> FFT -> Amplitude -> ... chain on its own shard with its own state, and no collective
sits on the data path.  Two things cross ranks:
  * the averaged spectrum of BASELINE config 5 -- one all-reduce(sum) of an F32[N] trace per
    reporting interval (256 KiB at N = 65536: latency-bound, so ONE collective per interval, never
    one per cycle);
  * spectrogram hit COUNTS when a single persistence display is wanted for all shards: the
    update `min(v + 0.02, 1)` applied count-times commutes with summing the integer counts, so an
    integer all-reduce keeps the merged display bit-exact (merge_hit_counts).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_batches(total_batches: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous split of `total_batches` rows over ranks: (start, count); sizes differ by <= 1."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("invalid rank / world size")
    base, extra = divmod(total_batches, world_size)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_average(trace: torch.Tensor) -> torch.Tensor:
    """Average an F32 trace (e.g. the lineplot's averaged spectrum) over all ranks, in place."""
    if _world() > 1:
        dist.all_reduce(trace, op=dist.ReduceOp.SUM)
        trace /= float(_world())
    return trace


def merge_hit_counts(counts: torch.Tensor) -> torch.Tensor:
    """Sum integer spectrogram hit counts [H, N] over ranks (exact), in place."""
    if _world() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


_COMM = None


def library_comm():
    """The library's own communicator (csrc/jst/comm.cc: RCCL behind the C ABI, no torch on the data path) for the
    torch.distributed world this process is part of: rank 0 creates the ncclUniqueId, torch.distributed -- the control
    plane that is already up -- broadcasts its 128 bytes, every rank calls jst_comm_init.  Created once per process."""
    global _COMM
    if _COMM is None:
        import cyberether_amd.jetstream as js
        world = _world()
        if world == 1:
            _COMM = js.Comm(0, 1, None)
        else:
            rank = dist.get_rank()
            ident = [js.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            _COMM = js.Comm(rank, world, ident[0])
    return _COMM


def _order_against_producer(stream: int, when: str) -> None:
    """Without an explicit stream the collective goes to the legacy null stream, which is NOT ordered against a runtime's
    stream (created hipStreamNonBlocking): the producer of the reduced tensor must have finished before the collective and
    the collective before the consumer's next cycle.  Pass the producing runtime's stream (Runtime.stream) to keep the
    exchange asynchronous; stream=0 keeps the round-3 contract instead -- device-wide synchronisation on both sides."""
    if stream == 0:
        torch.cuda.synchronize()


def merge_spectrogram_counts(counts_tensor, via_host: bool = False, stream: int = 0) -> None:
    """The exchange step of the exact multi-GPU spectrogram: sum the U32[H, N] hit counts a
    `spectrogram{merge=counts}` module wrote this cycle over all ranks, in place on the device -- jst_comm_allreduce (RCCL
    all-reduce of 4 MiB at H = 256, N = 4096) on `stream`; a `spectrogram_merge{batches = all ranks' batches}` module
    then applies them.  via_host: bounce through the host with torch.distributed (gloo dry runs where ranks share a
    device or have none).  stream: the stream of the runtime that produced the counts (ordered there, asynchronous); 0 =
    synchronise the device before and after (see _order_against_producer)."""
    if _world() == 1:
        return
    if via_host:
        host = torch.from_numpy(counts_tensor.numpy().astype(np.int32))
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        counts_tensor.copy_from(host.numpy().astype(np.uint32))
    else:
        _order_against_producer(stream, "before")
        library_comm().all_reduce(counts_tensor, "sum", stream=stream)
        _order_against_producer(stream, "after")


def average_trace(trace_tensor, stream: int = 0) -> None:
    """BASELINE config 5's averaged spectrum: the F32[N] trace of every rank becomes the mean over ranks, in place on the
    device (jst_comm_allreduce with average = 1), one collective per reporting interval.  stream: as merge_spectrogram_counts."""
    if _world() > 1:
        _order_against_producer(stream, "before")
        library_comm().all_reduce(trace_tensor, "sum", average=True, stream=stream)
        _order_against_producer(stream, "after")


def max_over_ranks(seconds: float, device: str = "cpu") -> float:
    """The bench contract's timing rule: the slowest rank defines the step time."""
    if _world() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def apply_hit_counts(bins: np.ndarray, counts: np.ndarray, decay: float) -> np.ndarray:
    """Host reference of how merged counts update a persistence display: bins*decay, then
    min(v + 0.02, 1) count-times (what spectrogram.hip does per workgroup tile)."""
    out = (bins.astype(np.float32) * np.float32(decay)).astype(np.float32)
    k = np.minimum(counts, 64)
    for _ in range(int(k.max()) if k.size else 0):
        hit = k > 0
        t = (out + np.float32(0.02)).astype(np.float32)
        out = np.where(hit, np.where(t > 1, np.float32(1), t), out).astype(np.float32)
        k = k - hit
    return out
