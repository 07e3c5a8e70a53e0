#!/usr/bin/env python3
"""How long does ONE launch of the fused spectrum kernel take per 1024 transforms when it carries B > 1024 of them?
(The persistent kernel's ramp / cold start / tail are paid once per launch: DESIGN.md section 4, "derived ceiling".)
ring_source(batches = B, 4096 samples, enough slots for > 256 MiB) -> spectrum_engine, provider fast / generic,
hipGraph, unit time by the runtime's event pairs and by wall clock."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.set_device(0)
    import cyberether_amd.jetstream as js
    n = 4096
    rng = np.random.default_rng(7)
    for provider in ("fast", "generic"):
        for batches in (1024, 2048, 4096, 8192, 16384):
            slots = max(2, (512 << 20) // (batches * n * 8))
            src = js.Module("ring_source", {"batches": batches, "samples": n, "slots": slots}, {}, "source")
            buf = src.output("buffer")
            x = (rng.standard_normal((1024, n, 2), dtype=np.float32) * np.float32(0.1)).view(np.complex64)[..., 0]
            for s in range(slots):
                buf.ring_select(s).copy_from(np.ascontiguousarray(np.tile(x, (batches // 1024, 1))))
            buf.ring_select(0)
            eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
            rt = js.Runtime([src] + eng.modules, graph=True, fuse=True, timing=False)
            rt.compute(4 * slots, sync=True)
            cycles = max(slots * 4, (64 * 1024) // batches * 4)
            cycles -= cycles % slots
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rt.compute(cycles, sync=False)
            rt.synchronize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / cycles
            per1024 = dt * 1e6 * 1024 / batches
            print(json.dumps({"provider": provider, "batches": batches, "slots": slots, "cycles": cycles,
                              "us_per_launch": dt * 1e6, "us_per_1024_transforms": per1024,
                              "frac_of_8TBps": 12.0 * 1024 * n / (per1024 * 1e-6) / 8e12,
                              "units": [u.split("(")[0] for u in rt.units]}), flush=True)
            rt.destroy()


if __name__ == "__main__":
    main()
