#!/usr/bin/env python3
"""The reference's Flowgraph / scheduler / Runtime on DeviceType::HIP (oracle/_ref/libref_jetstream_devhip.so) running N steady-state
cycles of ring_source -> spectrum_engine -> spectrogram, for a rocprofv3 --kernel-trace --memory-copy-trace run: between the fill of
the ring (host -> device, before the cycles) and the read-back (device -> host, after them) no copy may appear, whatever N is.
Modes: --mode sync (hand-off, one synchronous cycle per Flowgraph::compute) | modules (module by module).  Prints one JSON object."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=10)
    ap.add_argument("--mode", default="sync", choices=["sync", "modules"])
    args = ap.parse_args()
    import cyberether_amd.jetstream as js  # noqa: F401  (selects the device)
    from bench import synth_slot
    from oracle import ref_jetstream as rj
    rj.use_device_hip_library()
    rj.hip_runtime_configure(args.mode == "sync", 0)
    rows, n, h, slots = 1024, 4096, 256, 4
    rng = np.random.default_rng(1234)
    with rj.RefFlowgraph() as fg:
        assert fg.ring_source("src", rows, n, slots) == 0
        for s in range(slots):
            fg.ring_write("src", s, synth_slot(rng, s))
        assert fg.block("eng", "spectrum_engine", {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0}, {"buffer": "src:buffer"}, device="hip") == 0
        assert fg.block("spec", "spectrogram", {"height": h}, {"signal": "eng:buffer"}, device="hip") == 0
        assert fg.compute() == 0            # settle: window table upload, first eager cycle
        assert fg.compute_n(args.cycles) == 0
        units = rj.hip_runtime_units()
        out = np.array(fg.tensor("eng", "buffer"))
        bins = np.array(rj.hip_directory("spec-spectrogram", "state:frequencyBins"))
    print(json.dumps({"cycles": args.cycles, "mode": args.mode, "units": units.replace("\n", ", "), "output_peak": float(out.max()), "bins_sum": float(bins.sum())}))


if __name__ == "__main__":
    main()
