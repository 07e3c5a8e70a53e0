#!/usr/bin/env python3
"""BASELINE config 5 on N GPUs of one node: every rank runs its own 65536-point stream (Window -> FFT ->
Amplitude -> Range -> Lineplot average, 16 batches per cycle); once per reporting interval the averaged
PSD traces (F32[65536], 256 KiB) are all-reduced over RCCL/xGMI (cyberether_amd/distributed.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port 29531 tools/bench_c5_multi.py [--cycles 400] [--interval 25]

Prints one JSON line on rank 0: aggregate MS/s with the collective inside the timed region, and the
collective's own cost.  JST_BENCH_BACKEND=gloo is a dry run of the same script on a box with fewer GPUs
than ranks (ranks share devices, the trace takes a host round trip)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=400)
    ap.add_argument("--interval", type=int, default=25, help="cycles between PSD all-reduces")
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--fft", type=int, default=65536)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("JST_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    import cyberether_amd.jetstream as js
    from cyberether_amd import distributed as D
    js.set_device(local_rank)

    n, b = args.fft, args.batches
    rng = np.random.default_rng(1240 + rank)   # seeds 1240..1247
    t = np.arange(n) / 2.0e6
    x = (np.exp(2j * np.pi * (100.25 + rank) * 2.0e6 / n * t)[None, :] +
         1e-3 * (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n)))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0)
    lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
    rt = js.Runtime(eng.modules + [lp], graph=True, fuse=True)
    trace = lp.state("averagingBuffer")

    class _View:  # zero-copy torch view of the lineplot's device state
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(trace.data_ptr), False), "version": 2}
    dev_trace = None
    if backend != "gloo":
        dev_trace = torch.as_tensor(_View(), device="cuda")

    merged = None if backend == "gloo" else torch.empty(n, dtype=torch.float32, device="cuda")
    reduced = {}

    def reduce_psd() -> float:
        """The cross-rank mean of the averaged PSD goes into a SEPARATE trace: the lineplot's averagingBuffer is the
        state of each rank's own IIR recursion (lineplot/module_impl_native_cpu.cc:80-118) and must not be
        overwritten by the mean -- every rank keeps averaging ITS stream, the merged trace is what is displayed."""
        t0 = time.perf_counter()
        if backend == "gloo":
            host = torch.from_numpy(trace.numpy().copy())
            reduced["trace"] = D.allreduce_average(host)
        else:
            merged.copy_(dev_trace)          # 256 KiB device copy; the module's state stays this rank's own
            reduced["trace"] = D.allreduce_average(merged)
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    rt.compute(args.interval, sync=True)
    reduce_psd()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spent = 0.0
    done = 0
    while done < args.cycles:
        step = min(args.interval, args.cycles - done)
        rt.compute(step, sync=True)
        spent += reduce_psd()
        done += step
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, "cpu" if backend == "gloo" else "cuda")
    # the merged trace is the mean of the ranks' own (untouched) traces: check it on the way out
    own = torch.from_numpy(trace.numpy().copy()).to("cpu" if backend == "gloo" else "cuda")
    mean_check = D.allreduce_average(own.clone())
    merged_ok = bool(torch.allclose(reduced["trace"].cpu(), mean_check.cpu(), rtol=0, atol=1e-6))
    if rank == 0:
        print(json.dumps({
            "config": "C5: %d independent 65536-pt spectrum streams, PSD all-reduce every %d cycles" % (world, args.interval),
            "n_gpus": world, "cycles": args.cycles, "value": world * b * n * args.cycles / elapsed / 1e6,
            "unit": "MS/s", "us_per_cycle": elapsed / args.cycles * 1e6,
            "allreduce_ms_each": spent / max(1, (args.cycles + args.interval - 1) // args.interval) * 1e3,
            "merged_trace_is_mean_of_rank_traces": merged_ok, "backend": backend}), flush=True)
    rt.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
