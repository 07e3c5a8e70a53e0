#!/usr/bin/env python3
"""BASELINE config 5 as north_star writes it, on ONE GPU: all 8 streams resident -- 8 x 16 = 128 transforms of 65536 points per
cycle (Window -> FFT -> Amplitude -> Range -> Lineplot average) -- beside the one-stream form (16 transforms) the driver line
quotes.  Per cycle (one launch per unit and cycle) and cycle-batched on a resident ring.  28 B per sample."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cyberether_amd.jetstream as js  # noqa: E402


def timed(rt, cycles, warm):
    rt.compute(warm)
    t0 = time.perf_counter()
    rt.compute(cycles)
    return (time.perf_counter() - t0) / cycles


def main():
    # usage: bench_c5_streams.py [provider [batches]] -- provider generic (every float bit-identical, the default) or fast
    # (floats within 4e-7 of the reference CPU path: north_star allows 1e-5); batches 16 or 128 alone (one form per rocprofv3 trace)
    n = 65536
    provider = sys.argv[1] if len(sys.argv) > 1 else "generic"
    only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    form = sys.argv[3] if len(sys.argv) > 3 else ""  # "batched" | "per_cycle": that form alone
    out = {"provider": provider}
    for b, slots in ((16, 16), (128, 4)):
        if only and b != only:
            continue
        rng = np.random.default_rng(1240)
        t = np.arange(n)
        x = (np.exp(2j * np.pi * 1000.25 * t / n)[None, :] + 1e-3 * (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n)))).astype(np.complex64)
        rec = {}
        if form != "per_cycle":
            ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "iq")
            buf = ring.output("buffer")
            for sl in range(slots):
                buf.ring_select(sl).copy_from(np.roll(x, sl, axis=0))
            buf.ring_select(0)
            eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
            lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
            rt = js.Runtime([ring] + eng.modules + [lp], graph=True, fuse=True)
            dt = timed(rt, 40 * slots, 4 * slots)
            rec["batched"] = {"us_per_cycle": dt * 1e6, "frac_28B": 28.0 * b * n / dt / 8e12, "batched": bool(rt.batched), "units": [u.split("(")[0] for u in rt.units]}
            rt.destroy()
        if form != "batched":
            src = js.Tensor.from_numpy(x, batch=0, sample=1)
            eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
            lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
            rt = js.Runtime(eng.modules + [lp], graph=True, fuse=True, batch=False)
            dt1 = timed(rt, 200, 10)
            rec["per_cycle"] = {"us_per_cycle": dt1 * 1e6, "frac_28B": 28.0 * b * n / dt1 / 8e12}
            rt.destroy()
        out[f"{b}x{n}"] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
