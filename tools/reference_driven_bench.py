#!/usr/bin/env python3
"""The headline workload (BASELINE configs[1]: Window -> 4096-pt FFT -> Amplitude -> Range -> Spectrogram, 1024 batches of cf32
per cycle) DRIVEN BY THE REFERENCE: oracle/_ref/libref_jetstream_devhip.so is the reference's own core patched with
DeviceType::HIP (integration/device_hip/) -- its Flowgraph::blockCreate expands ring_source / spectrum_engine / spectrogram into
modules of (DeviceType::HIP, NATIVE), its synchronous scheduler orders and settles them and calls Runtime(HIP)::compute once per
cycle; the HIP runtime hands the library-only segment to one jst_runtime.  Three forms are timed, all device-resident:

  per_cycle_sync     the reference's contract as it stands: every Flowgraph::compute() returns with the cycle complete
                     (hipGraph replay of the fused cycle + one synchronise per cycle)
  per_cycle_async    opt-in (jetstream_hip_runtime_configure(1, 1)): every cycle is enqueued at once, nobody waits until flush()
  deferred_spans     opt-in for a resident ring (jetstream_hip_runtime_configure(1, R)): the runtime counts R cycles and runs
                     them as one cycle-batched span -- bench.py's headline form, reached from the reference's scheduler
  deferred_spans_sustained   the same with ONE flush per 16 spans: the scheduler's bookkeeping of span k + 1 (about 1.5 us of host
                     time per Flowgraph::compute()) overlaps the device's work on span k instead of preceding it
  module_by_module   hand-off disabled: one launch per module on the segment's stream (the CUDA runtime's shape)

and the deferred form is stamped from the reference's side: after 1 + 2R + 3 cycles the Spectrogram's bins (read from the
reference module's own state tensor in HBM) must equal the oracle's bit for bit and the engine's output must be within
north_star's 1e-5 of it (provider fast).  Prints one JSON object.  bench.py runs this in a process of its own (one build
of the reference per process) and carries the object as `reference_driven`."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_FFT, BATCHES, HEIGHT = 4096, 1024, 256
ENGINE = {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=20)
    ap.add_argument("--cycles", type=int, default=20, help="cycles per timed region (bench.py's --steps)")
    ap.add_argument("--min-time", type=float, default=0.25)
    ap.add_argument("--provider", default="fast")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import cyberether_amd.jetstream as js  # the product first: selects the device
    from bench import synth_slot
    from oracle import ref_jetstream as rj
    if not rj.device_hip_library_available():
        print(json.dumps({"available": False, "why": "oracle/_ref/libref_jetstream_devhip.so not built"}))
        return
    assert js.device_count() >= 1
    rj.use_device_hip_library()
    rng = np.random.default_rng(1234)
    ring = [synth_slot(rng, s) for s in range(args.slots)]

    def build(fg):
        assert fg.ring_source("src", BATCHES, N_FFT, args.slots, provider=args.provider) == 0
        for s in range(args.slots):
            fg.ring_write("src", s, ring[s])
        assert fg.block("eng", "spectrum_engine", ENGINE, {"buffer": "src:buffer"}, provider=args.provider, device="hip") == 0
        assert fg.block("spec", "spectrogram", {"height": HEIGHT}, {"signal": "eng:buffer"}, provider=args.provider, device="hip") == 0

    def timed(hand_off, defer: int, cycles: int = 0) -> dict:
        cycles = cycles or args.cycles
        rj.hip_runtime_configure(hand_off, defer)
        with rj.RefFlowgraph() as fg:
            build(fg)
            assert fg.compute() == 0                      # the settle cycle (static window chain, table uploads)
            assert fg.compute_n(2 * args.slots - 1) == 0  # prime: graph instantiation, span graphs
            rj.hip_runtime_flush()
            units = rj.hip_runtime_units()
            times = []
            total = 0.0
            while total < args.min_time or len(times) < 3:
                t0 = time.perf_counter()
                assert fg.compute_n(cycles) == 0
                rj.hip_runtime_flush()
                dt = time.perf_counter() - t0
                times.append(dt)
                total += dt
                if (-cycles) % args.slots:
                    assert fg.compute_n((-cycles) % args.slots) == 0   # untimed: back to the starting phase
                    rj.hip_runtime_flush()
            us = 1e6 * float(np.median(times)) / cycles
        return {"us_per_cycle": us, "MSps": BATCHES * N_FFT / us, "cycles_per_region": cycles, "regions": len(times),
                "units": units.replace("\n", ", ").rstrip(", ")}

    out = {"available": True, "workload": f"configs[1] through the reference's Flowgraph / scheduler / Runtime(HIP): ring_source[{args.slots} x "
                                         f"{BATCHES} x {N_FFT}] -> spectrum_engine -> spectrogram[{HEIGHT}], DeviceType::HIP, provider {args.provider}",
           "cycles_per_region": args.cycles,
           "per_cycle_sync": timed(True, 0), "per_cycle_sync_hipgraph": timed(3, 0), "per_cycle_async": timed(True, 1), "per_cycle_async_hipgraph": timed(3, 1), "deferred_spans": timed(True, args.slots),
           "deferred_spans_sustained": timed(True, args.slots, 16 * args.slots), "module_by_module": timed(False, 0)}

    if not args.no_parity:
        from oracle import oracle
        cycles = 1 + 2 * args.slots + 3
        rj.hip_runtime_configure(True, args.slots)
        with rj.RefFlowgraph() as fg:
            build(fg)
            assert fg.compute_n(cycles) == 0
            rj.hip_runtime_flush()
            got = np.array(fg.tensor("eng", "buffer"))
            bins = np.array(rj.hip_directory("spec-spectrogram", "state:frequencyBins")).reshape(-1)
        refs = [oracle.spectrum_chain(x, -100.0, 0.0)["range"] for x in ring]
        ref_bins = np.zeros(N_FFT * HEIGHT, np.float32)
        for c in range(cycles):
            oracle.spectrogram(ref_bins, refs[c % args.slots], HEIGHT)
        err = float(np.max(np.abs(got - refs[(cycles - 1) % args.slots])))
        out["parity"] = {"checked": True, "cycles": cycles, "form": "deferred_spans",
                         "spectrogram_words": int(bins.size), "spectrogram_bit_exact": bool(np.array_equal(bins.view(np.uint32), ref_bins.view(np.uint32))),
                         "output_max_abs_err": err, "output_within_1e-5": bool(err <= 1e-5), "output_rows": BATCHES}
    rj.hip_runtime_configure(True, 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
