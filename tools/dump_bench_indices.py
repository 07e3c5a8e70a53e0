#!/usr/bin/env python3
"""Writes the row indices the bench's fused kernel hands the Spectrogram (U8[slots][1024][4096], derived from the range
output exactly as the kernel does: (u32)(v * 256) where 1 <= v * 256 < 256, else 0) to a file, for the span-kernel
micro-benchmarks (tools/ubench/spec_span_timeline.hip <cycles> <file>)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.set_device(0)
    import cyberether_amd.jetstream as js
    out, slots = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
    src = js.Module("ring_source", {"batches": bench.BATCHES, "samples": bench.N_FFT, "slots": slots}, {}, "source")
    buf = src.output("buffer")
    rng = np.random.default_rng(1234)
    for s in range(slots):
        buf.ring_select(s).copy_from(bench.synth_slot(rng, s))
    buf.ring_select(0)
    eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider="generic")
    rt = js.Runtime([src] + eng.modules, fuse=True)
    with open(out, "wb") as f:
        for s in range(slots):
            rt.compute(1)
            v = eng.buffer.numpy() * np.float32(256.0)
            idx = np.where((v >= 1.0) & (v < 256.0), v, 0.0).astype(np.uint32).astype(np.uint8)
            f.write(idx.tobytes())
            if s == 0:
                col = idx[:, 2000].astype(np.int64)
                print("slot 0, column 2000: bins used", np.count_nonzero(np.bincount(col, minlength=256)),
                      "max count", np.bincount(col, minlength=256).max(), "zeros", int((col == 0).sum()))


if __name__ == "__main__":
    main()
