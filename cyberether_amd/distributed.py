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


_TORCH_OF = {"F32": (torch.float32, "<f4"), "U32": (torch.int32, "<i4"), "I32": (torch.int32, "<i4")}


def device_view(tensor) -> torch.Tensor:
    """Zero-copy torch view of a dense HIP jetstream Tensor (F32 / U32) for a collective: the collective then runs
    in place on the module's own HBM.  U32 hit counts travel as int32 (sums stay far below 2^31)."""
    if tensor.device != "hip":
        raise ValueError("device_view needs a HIP tensor")
    tdtype, typestr = _TORCH_OF[tensor.dtype]
    shape = tuple(int(v) for v in tensor.shape)

    class _View:
        __cuda_array_interface__ = {"shape": shape, "typestr": typestr,
                                    "data": (int(tensor.data_ptr) + int(tensor.offset) * 4, False), "version": 2}
    view = torch.as_tensor(_View(), device="cuda")
    view._jst_keep = tensor   # the jetstream tensor owns the memory
    return view


def merge_spectrogram_counts(counts_tensor, via_host: bool = False) -> None:
    """The exchange step of the exact multi-GPU spectrogram: sum the U32[H, N] hit counts a
    `spectrogram{merge=counts}` module wrote this cycle over all ranks, in place on the device (RCCL all-reduce of
    4 MiB at H = 256, N = 4096); a `spectrogram_merge{batches = all ranks' batches}` module then applies them.
    via_host: bounce through the host (gloo dry runs where ranks share a device)."""
    if _world() == 1:
        return
    if via_host:
        host = torch.from_numpy(counts_tensor.numpy().astype(np.int32))
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        counts_tensor.copy_from(host.numpy().astype(np.uint32))
    else:
        merge_hit_counts(device_view(counts_tensor))
        torch.cuda.synchronize()


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
