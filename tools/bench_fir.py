#!/usr/bin/env python3
"""BASELINE config 3 on provider fast (one FIR + /10 kernel): the MFMA form against the direct form (JST_FIR_DIRECT), same box, same
input; error of both against the bit-exact FFT overlap-add chain.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cyberether_amd.jetstream as js  # noqa: E402


def main():
    b, s, taps, sr, bw = 100, 159750, int(os.environ.get("TAPS", "251")), 20e6, 2e6
    rng = np.random.default_rng(1235)
    t = np.arange(s) / sr
    tones = (np.exp(2j * np.pi * 0.3e6 * t) + 0.5 * np.exp(2j * np.pi * 4.0e6 * t)).astype(np.complex64)
    x = tones[None, :] + (0.01 * (rng.standard_normal((b, s)) + 1j * rng.standard_normal((b, s)))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    exact = js.Filter(src, sr, bw, [0.0], taps, 1)
    rt = js.Runtime(exact.modules, graph=True, fuse=True)
    rt.compute(2)
    want = exact.buffer.numpy()
    rt.destroy()
    peak = float(np.max(np.abs(want)))
    out = {"taps": taps, "shape": [b, s]}
    for name, direct in (("direct", "1"), ("mfma", None), ("direct_again", "1"), ("mfma_again", None)):
        js.debug_set("JST_FIR_DIRECT", direct)
        blk = js.Filter(src, sr, bw, [0.0], taps, 1, provider="fast")
        rt = js.Runtime(blk.modules, graph=True, fuse=True)
        rt.compute(2)
        got = blk.buffer.numpy()
        err = float(np.max(np.abs(got - want)) / peak)
        rt.compute(10)
        t0 = time.perf_counter()
        rt.compute(200)
        dt = (time.perf_counter() - t0) / 200
        out[name] = {"us_per_cycle": dt * 1e6, "frac_8p8": 8.8 * b * s / dt / 8e12, "max_err_rel_peak": err, "units": [u.split("(")[0] for u in rt.units]}
        rt.destroy()
    js.debug_set("JST_FIR_DIRECT", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
