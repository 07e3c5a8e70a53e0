#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Jetstream backend.

Workload (BASELINE.json configs[1]): Window -> 4096-pt FFT -> Amplitude -> Range -> Spectrogram on
1024 batches of cf32 IQ per compute cycle, the whole cycle captured in a hipGraph.  One "step" =
one compute cycle over one batch tensor CF32[1024, 4096] that is ALREADY RESIDENT IN HBM: a ring
of `--slots` distinct batches (default 32 x 32 MiB = 1 GiB, four times the 256 MiB Infinity
Cache, so every step's input really comes from HBM).

Cycle batching (default; `--no-batch` and `alt_per_cycle_launch` are the other form): the runtime captures a ring period of R
(= slots) cycles into its graph anyway, and with all R slots resident it submits them as ONE launch per unit -- the persistent
fused kernel over R x 1024 transforms, the Spectrogram over the R index tensors with its state in registers -- instead of R
launches each (JST_RUNTIME_BATCH; R = 32 since round 4: same-box 12.78 -> 12.36 us per step against
R = 16, profiles/r04_experiments/s_ring_period_and_value_store_policy.log).  Every step is still one pass over one CF32[1024, 4096] batch: its
range output lands in its slot of the output ring, the Spectrogram state takes that cycle's decay and hit update, and the
parity leg checks both per cycle; what changes is that the kernel's ramp, cold start and tail are paid once per R steps.
`roofline` prices the launch that is timed: cycles_per_launch x 50.33 MB over its event-pair duration.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (the fused spectrum kernel): algorithmic bytes per launch
                  (12 B per complex sample: 8 B cf32 read + 4 B f32 write; a cycle-batched
                  launch carries cycles_per_launch x 1024 transforms) over its mean launch duration measured
                  with hipEvent pairs recorded on the runtime's own stream inside the timed region (every
                  sixteenth ring period is submitted eagerly between real event records).
  cpu_baseline -- the reference's CPU path timed on this host: dense C loops per stage (oracle/jst_oracle.c),
                  the FFT through the reference's OWN pocketfft (oracle/_ref, kind "reference"; the C
                  restatement, kind "port", only when that library is absent), nanobench-style like
                  src/benchmark.cc:100-106,175-186 (warm-up, epochs, median), on 1 core (the reference's
                  compute path is single-threaded, fft/module_impl_native_cpu.cc:1-2) plus an `all_cores`
                  figure from one independent replica per host core.
  parity       -- AFTER the timed region, outside it: the very runtime that was timed (same graphs, same ring data)
                  runs one more ring period cycle by cycle, then a whole-period graph replay and a tail; a sample of
                  rows of every slot's range output and the full spectrogram state are recomputed by the oracle
                  (the checker; the timed leg never touches it) and compared bit for bit.
  reference_driven -- the same workload with the REFERENCE's own Flowgraph, scheduler and Runtime in charge, device-resident on
                  DeviceType::HIP (integration/device_hip/ linked into oracle/_ref/libref_jetstream_devhip.so): synchronous
                  cycles, asynchronous cycles, deferred cycle-batched spans, module by module -- with a parity stamp taken
                  from the reference module's own state tensor (tools/reference_driven_bench.py).
  host_fed     -- the same chain fed from PINNED HOST memory: every cycle's batch is uploaded with
                  jst_tensor_copy_from_host_async on the library's side stream into the next ring slot while
                  the previous slot computes (the HBM replacement of the Soapy CircularBuffer hand-over,
                  soapy/module_impl_native_cpu.cc:39-60).  PCIe-bound by construction; never `value`.

Any --steps works: cycles that do not fill a ring period replay as captured span graphs (Runtime::launchSpan),
and a timed region shorter than 0.25 s is repeated (phase-aligned, primed once untimed) and averaged.
`--gpus N` without a torch.distributed.run environment re-executes itself under it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT = 4096
BATCHES = 1024
HEIGHT = 256
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 12.0    # 8 B cf32 in + 4 B f32 out (SURVEY 8d / DESIGN.md section 4)
STEP_BYTES_PER_SAMPLE = 14.0    # + the spectrogram's state read-modify-write, 2*4*N*H per cycle = 2 B/sample at B = 1024


def synth_slot(rng: np.random.Generator, slot: int) -> np.ndarray:
    """CF32[BATCHES, N_FFT]: row r = unit CW tone at bin 100.25 + r (+ slot) + AWGN sigma 1e-3."""
    n = np.arange(N_FFT, dtype=np.float64)
    bins = (100.25 + np.arange(BATCHES, dtype=np.float64) + slot) % N_FFT
    phase = 2.0 * np.pi * bins[:, None] * n[None, :] / N_FFT
    x = np.empty((BATCHES, N_FFT), np.complex64)
    x.real = np.cos(phase)
    x.imag = np.sin(phase)
    noise = rng.standard_normal((BATCHES, N_FFT, 2), dtype=np.float32) * np.float32(1e-3)
    x.real += noise[..., 0]
    x.imag += noise[..., 1]
    return x


def cpu_baseline() -> dict:
    """1 core: median of 15 epochs of >= 0.6 s (about 10 s); all cores: one replica process per host core,
    5 epochs of >= 0.4 s each, running at the same time (about 3 s), medians summed."""
    import subprocess
    from oracle import chain_bench
    one = chain_bench.run(rows=64, epoch_s=0.6, epochs=15)
    cores = os.cpu_count() or 1
    replicas = min(cores, 256)
    cmd = [sys.executable, "-m", "oracle.chain_bench", "--rows", "64", "--epoch-s", "0.4", "--epochs", "5"]
    procs = [subprocess.Popen(cmd + ["--seed", str(5000 + i)], cwd=ROOT, stdout=subprocess.PIPE, text=True,
                              env={**os.environ, "OMP_NUM_THREADS": "1"}) for i in range(replicas)]
    total = 0.0
    for pr in procs:
        out, _ = pr.communicate(timeout=300)
        try:
            total += json.loads(out.strip().splitlines()[-1])["samples_per_s"]
        except (ValueError, IndexError, KeyError):
            pass
    c0 = chain_bench.run_configs0(epoch_s=0.1, epochs=11)
    # The same chain through the REFERENCE'S OWN RUNTIME (oracle/_ref/libref_jetstream.so: its scheduler, native-CPU runtime,
    # spectrum_engine block and modules -- AutomaticIterator and all -- compiled in place; Flowgraph::compute() timed
    # nanobench-style in C).  When the library is there this is the baseline of record; the dense C loops around the
    # reference's pocketfft (what earlier rounds reported) ride along as `dense_port`.
    rr = chain_bench.run_reference_runtime(rows=64, epoch_s=0.6, epochs=9)
    dense = {"value": one["samples_per_s"] / 1e6, "unit": "MS/s", "cores": 1, "kind": one["kind"],
             "sample": f"64 batches x {N_FFT}-pt per pass through multiply/FFT/amplitude/range/spectrogram in dense C loops, "
                       f"FFT = {'the reference pocketfft (oracle/_ref)' if one['kind'] == 'reference' else 'C restatement'}, "
                       f"median of {one['epochs']} epochs of >= {one['epoch_s']} s (nanobench-style)",
             "epoch_rates_MSps": one["epoch_rates_MSps"]}
    head = dense if rr is None else {
        "value": rr["samples_per_s"] / 1e6, "unit": "MS/s", "cores": 1, "kind": "reference-runtime",
        "sample": f"64 batches x {N_FFT}-pt per Flowgraph::compute() of the REFERENCE compiled in place (oracle/_ref/"
                  f"libref_jetstream.so): source -> spectrum_engine block (cast, window, invert, reshape, multiply, fft, "
                  f"amplitude, range: its own native-CPU modules) -> spectrogram block, inside its synchronous scheduler; "
                  f"median of {rr['epochs']} epochs of >= {rr['epoch_s']} s (src/benchmark.cc:100-106,175-186 shape)",
        "epoch_rates_MSps": rr["epoch_rates_MSps"], "dense_port": dense}
    return {**head,
            "configs0": {"value": c0["samples_per_s"] / 1e6, "unit": "MS/s", "us_per_op": c0["us_per_op"], "cores": 1,
                         "kind": c0["kind"],
                         "sample": f"BASELINE configs[0]: 1 batch x {N_FFT}-pt CW tone, FFT -> Amplitude per op, median of "
                                   f"{c0['epochs']} epochs of >= {c0['epoch_s']} s (src/benchmark.cc:100-106,175-186 shape)"},
            "all_cores": {"value": total / 1e6, "unit": "MS/s", "cores": replicas,
                          "how": "one independent replica process per host core (the dense form), concurrently, medians summed"}}


def other_configs(js, budget_s: float = 45.0) -> list:
    """BASELINE.json configs[2..4] as secondary lines of the SAME driver-run command: each on the literal
    SURVEY section 8(d) input, device resident, graph captured; per config the time per compute cycle, the whole-chain HBM
    roofline fraction on section 8(d)'s algorithmic bytes, and a parity stamp against the CPU oracle computed on the very
    tensors that were timed (the oracle is the checker, outside every timed region).  Time-boxed: a config that does not
    fit what is left of `budget_s` is reported as skipped."""
    import torch
    from oracle import oracle
    t_start = time.perf_counter()
    out = []

    def timed(rt, cycles, warmup):
        rt.compute(warmup, sync=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rt.compute(cycles, sync=False)
        rt.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / cycles

    def roof(bytes_per_sample, samples, dt):
        a = bytes_per_sample * samples / dt / 1e9
        return {"bound": "hbm", "bytes_per_sample": bytes_per_sample, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": a / HBM_PEAK_GBS}

    def same(a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        return a.shape == b.shape and bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))

    def guarded(name, fn):
        if time.perf_counter() - t_start > budget_s:
            out.append({"config": name, "skipped": "time box"})
            return
        try:
            t0 = time.perf_counter()
            rec = fn()
            rec["config"] = name
            rec["seconds"] = round(time.perf_counter() - t0, 2)
            out.append(rec)
        except Exception as exc:  # the headline must not depend on a secondary line
            out.append({"config": name, "error": repr(exc)})

    def c3():  # 251-tap band-pass + /10 on CF32[100, 159750] (16 MS per cycle), FFT overlap-add, two distinct batches
        b, s, taps, sr, bw = 100, 159750, 251, 20e6, 2e6
        rng = np.random.default_rng(1235)
        t = np.arange(s) / sr
        tones = (np.exp(2j * np.pi * 0.3e6 * t) + 0.5 * np.exp(2j * np.pi * 4.0e6 * t)).astype(np.complex64)
        xs = [(tones[None, :] + (0.01 * (rng.standard_normal((b, s)) + 1j * rng.standard_normal((b, s)))).astype(np.complex64))
              for _ in range(2)]
        src = js.Tensor.from_numpy(xs[0], batch=0, sample=1)
        blk = js.Filter(src, sr, bw, [0.0], taps, 1)
        rt = js.Runtime(blk.modules, graph=True, fuse=True)
        # parity: cycle 1 on batch 0, cycle 2 on batch 1 (overlap state crosses the compute boundary); rows 0..3 of both
        state, ok = {}, True
        for c, x in enumerate(xs):
            src.copy_from(x)
            rt.compute(1)
            got = blk.buffer.numpy()
            want = oracle.filter_block(x[:4], blk.plan, sr, bw, [0.0], taps, state)
            ok &= same(got[:4], want)
            state = {}
            oracle.filter_block(x[95:], blk.plan, sr, bw, [0.0], taps, state)   # the tail row 99 hands to the next cycle
            state = {"prev": state["prev"]}
        dt = timed(rt, 20, 3)
        units = rt.units
        exact_last = blk.buffer.numpy()          # the last timed cycle ran on batch 1 with batch 1's own state behind it
        rt.destroy()
        rec = {"ms_per_cycle": dt * 1e3, "MS_per_s_in": b * s / dt / 1e6, "roofline": roof(32.0, b * s, dt),
               # the same time on SURVEY 8(d)'s TIME-DOMAIN accounting of this config (8 B read + 0.8 B written per input sample)
               "roofline_on_8p8_bytes": roof(8.8, b * s, dt)["frac"],
               "units": [u.split("(")[0] for u in units],
               "parity": {"checked": True, "bit_exact": ok, "rows_compared": 4, "rows": b,
                          "what": "rows 0..3 of 100 (4 % of the rows; the full-size GPU test compares rows 0..7 and 92..99) of "
                                  "two consecutive cycles on distinct batches (carried overlap state) vs oracle.filter_block"}}
        # BASELINE's own wording of this config -- "LDS tap stencil": provider fast of the Filter block = ONE direct-form
        # polyphase FIR + /10 kernel (fir.hip: fir_decimate_kernel).  Same taps, floats within north_star's tolerance:
        # stamped against the bit-exact chain's output of the same input AND against the oracle on rows 0..3.
        blk = js.Filter(src, sr, bw, [0.0], taps, 1, provider="fast")
        rt = js.Runtime(blk.modules, graph=True, fuse=True)
        src.copy_from(xs[1])
        rt.compute(2)                            # same input twice: the history equals the chain's overlap state
        fast = blk.buffer.numpy()
        peak = float(np.max(np.abs(exact_last)))
        err_chain = float(np.max(np.abs(fast[1:] - exact_last[1:])) / peak)
        dtf = timed(rt, 40, 4)
        unitsf = rt.units
        rt.destroy()
        rec["fast"] = {"ms_per_cycle": dtf * 1e3, "MS_per_s_in": b * s / dtf / 1e6, "roofline": roof(8.8, b * s, dtf),
                       "units": [u.split("(")[0] for u in unitsf],
                       "parity": {"checked": True, "max_err_rel_peak": err_chain, "within_1e-5": bool(err_chain <= 1e-5),
                                  "what": "rows 1..99 of the direct-form kernel's output vs the bit-exact FFT overlap-add chain's "
                                          "(itself stamped against the oracle above) on the same batch, |error| / peak magnitude; "
                                          "row 0 differs by construction (the chain's first row starts from the previous "
                                          "CYCLE's tail, the stencil's from its own history)"}}
        return rec

    def c4():  # 20 MS/s -> Filter(/100) -> FM wide 75us -> Decimator(/4), one stereo lane, 10 batches per cycle
        b, s, taps, sr, bw = 10, 202400, 101, 20e6, 200e3
        tt = np.arange(b * s) / sr
        audio = 0.45 * np.sin(2 * np.pi * 1e3 * tt) + 0.1 * np.sin(2 * np.pi * 19e3 * tt)
        x = np.exp(2j * np.pi * 75e3 * np.cumsum(audio) / sr).astype(np.complex64).reshape(b, s)
        src = js.Tensor.from_numpy(x, batch=0, sample=1)
        filt = js.Filter(src, sr, bw, [0.0], taps, 1)
        sq = js.Module("squeeze_dims", {"axis": 1}, {"buffer": filt.buffer}, "squeeze_head")
        iq = sq.output("buffer").set_axes(batch=0, sample=1)
        fm = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3}, {"signal": iq}, "fm")
        dec = js.Decimator(fm.output("signal"), 4)
        rt = js.Runtime(filt.modules + [sq, fm] + dec.modules, graph=True, fuse=True)
        rt.compute(1)
        base = oracle.filter_block(x, filt.plan, sr, bw, [0.0], taps, {})
        lane = oracle.FmLane("wide", "75us", 200e3)
        stereo = np.asarray(lane(np.ascontiguousarray(base[:, 0, :])), np.float32).reshape(b, 2024, 2)
        want = oracle.arithmetic_add(np.ascontiguousarray(stereo.reshape(b, 506, 4, 2)), 2).reshape(b, 506, 2)
        ok = same(filt.buffer.numpy(), base) and same(fm.output("signal").numpy(), stereo) and same(dec.buffer.numpy(), want)
        dt = timed(rt, 12, 2)
        rt.destroy()
        rec = {"ms_per_cycle": dt * 1e3, "MS_per_s_in": b * s / dt / 1e6, "x_realtime": (b * s / sr) / dt,
               "roofline": roof(8.0 + 0.08 * 32.0, b * s, dt),
               "note": "the stereo decode is a chain of serial recurrences per station: latency bound, not HBM bound -- one "
                       "station is one workgroup; the lever is stations (below)",
               "parity": {"checked": True, "bit_exact": ok, "what": "first cycle: filter output, stereo decode and /4 "
                          "integrate-and-dump vs the oracle (whole tensors)"}}
        # the same decoder on 64 stations at once (lanes are independent workgroups): what a multi-channel receiver runs
        lanes, nb, ns = 64, 10, 2024
        tl = np.arange(nb * ns) / 200e3
        one = np.exp(2j * np.pi * 75e3 * np.cumsum(0.45 * np.sin(2 * np.pi * 1e3 * tl) + 0.1 * np.sin(2 * np.pi * 19e3 * tl)) / 200e3)
        xl = np.stack([np.roll(one, 37 * l) for l in range(lanes)], axis=0).astype(np.complex64)
        xl = np.ascontiguousarray(xl.reshape(lanes, nb, ns).transpose(1, 0, 2))                 # [batches, stations, samples]
        tsr = js.Tensor.from_numpy(xl, batch=0, sample=2)
        fm64 = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3}, {"signal": tsr}, "fm64")
        rt = js.Runtime([fm64], graph=True)
        rt.compute(1)
        got = fm64.output("signal").numpy()
        ok64 = True
        for l in (0, 1, 31, 63):                 # four of the 64 stations against the oracle's serial lane, whole first cycle
            ln = oracle.FmLane("wide", "75us", 200e3)
            ok64 &= same(got[:, l], np.asarray(ln(np.ascontiguousarray(xl[:, l, :])), np.float32).reshape(nb, ns, 2))
        dt64 = timed(rt, 12, 2)
        rt.destroy()
        rec["stations_64"] = {"ms_per_cycle": dt64 * 1e3, "stations_x_realtime": lanes * (nb * ns / 200e3) / dt64,
                              "what": "fm{wide, 75us} on CF32[10, 64, 2024] at 200 kS/s: 64 stations decoded side by side",
                              "parity": {"checked": True, "bit_exact": bool(ok64),
                                         "what": "stations 0, 1, 31, 63 of the first cycle vs oracle.FmLane (whole lanes)"}}
        return rec

    def c5():  # one stream of config 5: Window -> 65536-pt FFT -> Amplitude -> Range -> Lineplot average, 16 batches
        n, b, slots = 65536, 16, 16
        rng = np.random.default_rng(1240)
        t = np.arange(n)
        x = (np.exp(2j * np.pi * 1000.25 * t / n)[None, :] + 1e-3 * (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n)))).astype(np.complex64)
        # the source is what the reference's sources are, a ring of resident batches (soapy/module_impl.cc's circular
        # buffer; a file / replay source): the runtime's default (graph + fuse => cycle batching) takes the cycles of a ring
        # period as one columns / blocks / lineplot launch each
        ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "iq")
        buf = ring.output("buffer")
        for sl in range(slots):
            buf.ring_select(sl).copy_from(np.roll(x, sl, axis=0))
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0)
        lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
        rt = js.Runtime([ring] + eng.modules + [lp], graph=True, fuse=True)
        rt.compute(1)
        ok = same(eng.buffer.numpy(), oracle.spectrum_chain(x, -100.0, 0.0)["range"])
        batched = bool(rt.batched)
        dt = timed(rt, 320, 48)
        # the stamp of the form that was TIMED: one more whole ring period as cycle-batched span launches, then every slot of
        # the output ring against the oracle (slot s holds the spectrum of np.roll(x, s): a row permutation of the first)
        rt.compute((-(1 + 320 + 48)) % slots + slots)
        first = oracle.spectrum_chain(x, -100.0, 0.0)["range"]
        ok_span = all(same(eng.buffer.ring_select(sl).numpy(), np.roll(first, sl, axis=0)) for sl in range(slots)) if batched else None
        rt.destroy()
        rec = {"us_per_cycle": dt * 1e6, "MS_per_s": b * n / dt / 1e6, "roofline": roof(28.0, b * n, dt),
               "source": f"resident ring of {slots} slots, runtime defaults", "cycle_batched": batched,
               "parity": {"checked": True, "bit_exact": bool(ok and ok_span is not False), "first_cycle_bit_exact": ok,
                          "batched_span_bit_exact": ok_span, "slots_compared": slots if batched else 0,
                          "what": "range output of the first (eager) cycle AND of every slot of the output ring after a whole "
                                  "ring period of cycle-batched span launches -- the form that is timed -- vs "
                                  "oracle.spectrum_chain (16 x 65536 per slot)"}}
        # one launch per unit and cycle on a plain tensor (rounds 1-3 quoted this form)
        src = js.Tensor.from_numpy(x, batch=0, sample=1)
        eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0)
        lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
        rt = js.Runtime(eng.modules + [lp], graph=True, fuse=True, batch=False)
        rt.compute(1)
        ok1 = same(eng.buffer.numpy(), oracle.spectrum_chain(x, -100.0, 0.0)["range"])
        dt1 = timed(rt, 100, 10)
        rt.destroy()
        rec["launch_per_cycle"] = {"us_per_cycle": dt1 * 1e6, "roofline_frac": roof(28.0, b * n, dt1)["frac"], "bit_exact": ok1}
        # north_star's own reading of the config -- EIGHT streams -- on one GPU: 8 x 16 = 128 transforms per cycle fill the CUs
        # (16 transforms leave half of them idle and three launch floors dominate); one launch per unit and cycle
        b8 = 8 * b
        x8 = np.concatenate([np.roll(x, 3 * k, axis=1) * np.float32(1.0 + 0.125 * k) for k in range(8)], axis=0)
        src8 = js.Tensor.from_numpy(x8, batch=0, sample=1)
        eng8 = js.SpectrumEngine(src8, enable_scale=True, range_min=-100.0, range_max=0.0)
        lp8 = js.Module("lineplot", {"averaging": 8}, {"signal": eng8.buffer}, "psd")
        rt8 = js.Runtime(eng8.modules + [lp8], graph=True, fuse=True, batch=False)
        rt8.compute(1)
        rows8 = np.r_[0:4, b8 - 4:b8]
        ok8 = same(eng8.buffer.numpy()[rows8], oracle.spectrum_chain(x8[rows8], -100.0, 0.0)["range"])
        dt8 = timed(rt8, 60, 6)
        rt8.destroy()
        rec["streams_8"] = {"transforms_per_cycle": b8, "us_per_cycle": dt8 * 1e6, "MS_per_s": b8 * n / dt8 / 1e6,
                            "roofline_frac": roof(28.0, b8 * n, dt8)["frac"], "bit_exact_rows_0_3_and_124_127": ok8,
                            "what": "all 8 streams of configs[4] resident on ONE GPU: CF32[128, 65536] per cycle, one launch per unit and cycle"}
        # round 6: provider fast on the tiled path (the headline's provider: floats within 4e-7 of the reference CPU path, north_star
        # allows 1e-5; the lean epilogue instead of the libm restatement) -- one stream cycle-batched, 8 streams per cycle and
        # cycle-batched (a ring of 4 slots: 512 transforms per span launch, the persistent blocks kernel)
        fast = {}
        ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "iq")
        buf = ring.output("buffer")
        for sl in range(slots):
            buf.ring_select(sl).copy_from(np.roll(x, sl, axis=0))
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider="fast")
        lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
        rt = js.Runtime([ring] + eng.modules + [lp], graph=True, fuse=True)
        dtf = timed(rt, 320, 48)
        rt.compute((-(320 + 48)) % slots + slots)
        errf = max(float(np.max(np.abs(eng.buffer.ring_select(sl).numpy() - np.roll(first, sl, axis=0)))) for sl in range(slots))
        rt.destroy()
        fast["one_stream_batched"] = {"us_per_cycle": dtf * 1e6, "roofline_frac": roof(28.0, b * n, dtf)["frac"],
                                      "max_abs_err_vs_oracle_all_slots": errf, "within_1e-5": bool(errf <= 1e-5)}
        eng8 = js.SpectrumEngine(src8, enable_scale=True, range_min=-100.0, range_max=0.0, provider="fast")
        lp8 = js.Module("lineplot", {"averaging": 8}, {"signal": eng8.buffer}, "psd")
        rt8 = js.Runtime(eng8.modules + [lp8], graph=True, fuse=True, batch=False)
        rt8.compute(1)
        err8 = float(np.max(np.abs(eng8.buffer.numpy()[rows8] - oracle.spectrum_chain(x8[rows8], -100.0, 0.0)["range"])))
        dt8f = timed(rt8, 60, 6)
        rt8.destroy()
        fast["streams_8"] = {"us_per_cycle": dt8f * 1e6, "roofline_frac": roof(28.0, b8 * n, dt8f)["frac"],
                             "max_abs_err_vs_oracle_rows_0_3_and_124_127": err8, "within_1e-5": bool(err8 <= 1e-5)}
        ring8 = js.Module("ring_source", {"batches": b8, "samples": n, "slots": 4}, {}, "iq8")
        buf8 = ring8.output("buffer")
        for sl in range(4):
            buf8.ring_select(sl).copy_from(np.roll(x8, sl, axis=0))
        buf8.ring_select(0)
        eng8 = js.SpectrumEngine(buf8, enable_scale=True, range_min=-100.0, range_max=0.0, provider="fast")
        lp8 = js.Module("lineplot", {"averaging": 8}, {"signal": eng8.buffer}, "psd")
        rt8 = js.Runtime([ring8] + eng8.modules + [lp8], graph=True, fuse=True)
        dt8b = timed(rt8, 80, 12)
        batched8 = bool(rt8.batched)
        rt8.destroy()
        fast["streams_8_batched"] = {"us_per_cycle": dt8b * 1e6, "roofline_frac": roof(28.0, b8 * n, dt8b)["frac"], "cycle_batched": batched8,
                                     "what": "a resident ring of 4 slots x CF32[128, 65536]: 512 transforms per span launch"}
        rec["provider_fast"] = fast
        return rec

    guarded("configs[2]: 251-tap FIR (FFT overlap-add) + /10 on CF32[100,159750] (16 MS per cycle)", c3)
    guarded("configs[3]: WBFM 20 MS/s -> Filter(/100) -> FM wide 75us -> Decimator(/4)", c4)
    guarded("configs[4] per GPU: Window -> 65536-pt FFT -> Amplitude -> Range -> Lineplot average, 16 batches", c5)
    return out


def reference_driven(slots: int, steps: int, provider: str) -> dict:
    """The headline workload driven by the REFERENCE's own Flowgraph / scheduler / Runtime on DeviceType::HIP (the patched core of
    integration/device_hip/ linked with the library: oracle/_ref/libref_jetstream_devhip.so), in a process of its own (one build
    of the reference per process; the cpu_baseline leg loads the unpatched one).  tools/reference_driven_bench.py has the forms."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_jetstream_devhip.so")):
        return {"available": False, "why": "oracle/_ref/libref_jetstream_devhip.so not built (the reference tree was absent at build time)"}
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reference_driven_bench.py"), "--slots", str(slots),
                              "--cycles", str(steps), "--provider", provider], cwd=ROOT, capture_output=True, text=True, timeout=240)
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as exc:  # the headline must not depend on it
        return {"available": False, "error": repr(exc)}


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (the workload actually run -- it includes the Window and the
    Range stage the Spectrogram needs -- is spelled out in config.workload)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "MS/s complex IQ through FFT->Amplitude->Spectrogram @4096-pt; HBM GB/s %peak"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--slots", type=int, default=0,
                    help="ring slots = cycles of a ring period (0 = the run's steps between 16 and 32 -- the driver's --steps 20: 20: a timed region "
                         "holds at least one whole period, which is what the kernel's event pair brackets)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fuse", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timing", action="store_true", help="no hipEvent nodes in the graph")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the informational alt_provider / alt_combined / host_fed measurements (profiling runs: "
                         "one kernel variant, one launch pattern per trace)")
    ap.add_argument("--combine", action="store_true",
                    help="JST_RUNTIME_COMBINE: the spectrogram of cycle k - 1 rides on the fused spectrum launch of cycle k "
                         "(one kernel per cycle); default off: reported as alt_combined beside the headline")
    ap.add_argument("--no-batch", action="store_true",
                    help="one launch per unit and CYCLE (the round-2 form) instead of cycle batching (JST_RUNTIME_BATCH: the "
                         "cycles of a captured ring period run as one launch per unit -- the persistent fused kernel over all "
                         "resident slots, the Spectrogram over their index tensors); the per-cycle form is measured and "
                         "reported beside the headline either way (alt_per_cycle_launch)")
    ap.add_argument("--pipeline", action="store_true",
                    help="run the spectrogram as its own graph on a second stream, one ring period behind "
                         "the spectrum graph (two hardware queues: +6 %% throughput, the spectrum kernel "
                         "itself stretches ~5 %% while it shares the CUs)")
    ap.add_argument("--provider", default="fast", choices=["generic", "fast"],
                    help="amplitude/range arithmetic.  fast (the default since round 3): what north_star specifies -- "
                         "integer bin work bit-exact (the fused kernel's bin guard, proven on every input power), float "
                         "spectra within 4e-7 absolute of the reference CPU path (north_star allows 1e-5) -- through the "
                         "hardware transcendentals; generic: every float bit-identical to the reference CPU path (glibc "
                         "2.35 libm restated).  The other provider is measured too and reported as alt_provider.")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the pinned-host -> async H2D -> chain measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary lines for BASELINE configs[2..4]")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the post-measurement parity leg (profiling runs: nothing but the timed workload in the trace)")
    ap.add_argument("--min-time", type=float, default=0.25,
                    help="repeat the K-step timed region until this many seconds have been timed (0: once)")
    args = ap.parse_args()
    if args.slots <= 0:
        args.slots = min(32, max(16, args.steps))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
        # (one rank per GPU over RCCL); 127.0.0.1 because the container hostname may not resolve.
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    # ONE JSON line on stdout, nothing else: native libraries write there too (RCCL prints a five-line version banner through
    # C stdio when a communicator is created -- flushed at exit, BEHIND the JSON line).  File descriptor 1 points at stderr for
    # the whole run; the line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # JST_BENCH_BACKEND=gloo: dry run of the multi-rank path on a box with fewer GPUs than ranks (ranks
    # share devices, control plane on CPU tensors); the driver's runs use the default, RCCL.
    backend = os.environ.get("JST_BENCH_BACKEND", "nccl")
    if world > max(torch.cuda.device_count(), 1) and "JST_BENCH_BACKEND" not in os.environ:
        backend = "gloo"  # fewer GPUs than ranks (a dry run on a small box): ranks share devices
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)

    import cyberether_amd.jetstream as js  # fails loudly if the HIP library is missing
    js.set_device(local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def measure(provider: str, seed_offset: int = 0, pipeline: bool = args.pipeline, combine: bool = args.combine,
                batch: bool = not args.no_batch):
        """Builds ring_source -> spectrum_engine -> spectrogram with the given amplitude/range
        provider, runs W untimed + K timed steps; returns (runtime, elapsed seconds over ranks)."""
        source = js.Module("ring_source", {"batches": BATCHES, "samples": N_FFT, "slots": args.slots},
                           {}, "source")
        buf = source.output("buffer")
        rng = np.random.default_rng(1234 + rank + seed_offset)
        for s in range(args.slots):  # independent IQ per rank and slot, resident before timing
            buf.ring_select(s).copy_from(synth_slot(rng, s))
        buf.ring_select(0)
        engine = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0,
                                   provider=provider)
        spectrogram = js.Module("spectrogram", {"height": HEIGHT}, {"signal": engine.buffer},
                                "spectrogram")
        rt = js.Runtime([source] + engine.modules + [spectrogram], graph=not args.no_graph,
                        fuse=not args.no_fuse, timing=not args.no_timing,
                        pipeline=pipeline, combine=combine and not pipeline,
                        batch=batch and not pipeline and not combine and not args.no_graph and not args.no_fuse)
        rt._keep = (source, engine, spectrogram)  # module handles must outlive the runtime
        rt._seed = 1234 + rank + seed_offset
        # Initialisation, not measurement: the first replays of a freshly instantiated hipGraph carry its
        # one-time upload (milliseconds inside the first in-graph kernel's event pair), and a timed region
        # that starts off a period boundary runs its first cycles eagerly.  Two periods prime the graph;
        # after the W warmup steps a few more untimed steps (< one period) realign the cycle counter.
        period = max(rt.period, 1)
        rt.compute(2 * period, sync=True)
        rt.compute(args.warmup, sync=True)
        rt.compute((-args.warmup) % period, sync=True)
        # Prime: one untimed K-step region from the phase every timed region will start at (captures the span graph
        # of K mod period cycles, if any), then back to that phase.
        rt.compute(args.steps, sync=True)
        rt.compute((-args.steps) % period, sync=True)
        rt.reset_timing()

        sync_in_compute = os.environ.get("JST_BENCH_SYNC_IN_COMPUTE") == "1"

        def region() -> float:
            """EXACTLY K steps bracketed by barrier + synchronize on both sides; max over ranks."""
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            # submit, then ONE device-wide wait: torch.cuda.synchronize() covers the runtime's stream too (a second
            # hipStreamSynchronize inside compute() cost ~0.1 us per step of a 20-step region; JST_BENCH_SYNC_IN_COMPUTE=1: A/B)
            rt.compute(args.steps, sync=sync_in_compute)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rt.synchronize()   # untimed: the unit timers of the region's eager cycles are harvested here
            barrier()
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        times = [region()]
        # A region shorter than --min-time is repeated (every rank takes the same decision: times are the max over
        # ranks) from the same ring phase and the mean is reported.
        repeats = 1
        if args.min_time > 0 and times[0] < args.min_time:
            repeats = min(int(args.min_time / max(times[0], 1e-6)) + 1, 2000)
        for _ in range(repeats - 1):
            rt.compute((-args.steps) % period, sync=True)  # untimed: back to the starting phase
            times.append(region())
        elapsed = sum(times) / len(times)
        measure.repeats = len(times)
        measure.spread = (min(times), max(times))
        srt = sorted(times)
        measure.percentiles = [srt[int(q * (len(srt) - 1))] for q in (0.1, 0.5, 0.9)]
        return rt, elapsed

    def host_fed(provider: str, seconds: float = 0.3) -> dict:
        """The chain fed from the host through the live ring_source's PRODUCER interface (jst_ring_acquire / _commit /
        _push: the HBM replacement of the Soapy thread's CircularBuffer, soapy/module_impl.cc:375-399 +
        module_impl_native_cpu.cc:39-60).  The library assembles batches in pinned staging memory, uploads each on its
        own stream into the next ring slot while earlier slots compute, and orders uploads against the cycles that
        read the slots with per-slot events -- no synchronise in this loop.  Per sample format (cf32 / ci16 / ci8: 8 / 4
        / 2 bytes per sample over PCIe, the integer formats cast inside the fused kernel's first load):
          zero_copy  -- the producer owns the staging memory (a driver's readStream writing into it): acquire + commit,
                        no CPU copy; PCIe-bound
          push_8192  -- jst_ring_push of 8192-sample chunks out of pageable memory, the reference's Soapy loop
                        verbatim (the loop in native code, jst_probe_ring_push_chunks): bound by one core's copy into
                        the pinned staging memory (non-temporal stores since round 4)
        PCIe-inclusive; never `value`."""
        import ctypes as C
        slots = args.slots
        batch = BATCHES * N_FFT
        fmt = {"cf32": ("CF32", 8, None), "ci16": ("CI16", 4, np.int16), "ci8": ("CI8", 2, np.int8)}
        result = {"unit": "MS/s", "ring_slots": slots, "north_star_target_MSps": 2000.0,
                  "how": "live ring_source producer API (pinned staging -> async H2D on the source's upload stream -> "
                         "ring slot; per-slot events order uploads and cycles); PCIe-inclusive, never `value`"}
        for key, (dtype, bytes_per, np_t) in fmt.items():
            source = js.Module("ring_source", {"batches": BATCHES, "samples": N_FFT, "slots": slots, "live": True,
                                               "dtype": dtype}, {}, "source")
            buf = source.output("buffer")
            mods = [source]
            # the spectrum_engine block starts with its own cast (spectrum_engine/block_impl.cc:120-217): raw integer
            # samples go straight in, and the fusion planner folds that cast into the transform's first load
            engine = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
            spectrogram = js.Module("spectrogram", {"height": HEIGHT}, {"signal": engine.buffer}, "spectrogram")
            rt = js.Runtime(mods + engine.modules + [spectrogram], graph=False, fuse=not args.no_fuse, timing=False)
            rng = np.random.default_rng(99 + rank)
            x = synth_slot(rng, 0)
            if np_t is None:
                host = np.ascontiguousarray(x)
            else:
                full = float(np.iinfo(np_t).max)
                host = np.empty((BATCHES, N_FFT, 2), np_t)
                host[..., 0] = np.clip(np.round(x.real * (full * 0.5)), -full, full)
                host[..., 1] = np.clip(np.round(x.imag * (full * 0.5)), -full, full)
            nbytes = batch * bytes_per

            def lap(copy: bool) -> None:
                addr, room = source.ring_acquire()      # blocks while the staging buffer's last upload is in flight
                if copy:
                    C.memmove(addr, host.ctypes.data, nbytes)
                source.ring_commit(batch)
                rt.compute(1, sync=False)

            for _ in range(8):      # fills the four staging buffers with samples, settles the static units
                lap(True)
            rt.synchronize()
            torch.cuda.synchronize()
            cycles, t0 = 0, time.perf_counter()
            while True:
                for _ in range(slots):
                    lap(False)      # zero copy: the staging memory already holds a batch (a driver would have written it)
                cycles += slots
                if time.perf_counter() - t0 >= seconds:
                    break
            rt.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            zero = cycles * batch / dt
            # the reference's producer loop: 8192-sample pushes out of ordinary memory
            flat = host.reshape(-1, 2) if np_t is not None else host.reshape(-1)
            pushed, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds / 2:
                source.ring_push_chunks(flat, 8192)   # 512 pushes of 8192 samples, the loop itself in native code
                rt.compute(1, sync=False)
                pushed += 1
            rt.synchronize()
            torch.cuda.synchronize()
            push_rate = pushed * batch / (time.perf_counter() - t0)
            result[key] = {"zero_copy": zero / 1e6, "zero_copy_pcie_GBps": zero * bytes_per / 1e9,
                           "zero_copy_ms_per_step": dt / cycles * 1e3, "push_8192": push_rate / 1e6,
                           "vs_target": zero / 2.0e9, "overflows": source.ring_overflows,
                           "units": [u.split("(")[0] + ("(" + u.split("(")[1].split("+")[0] + "+..)" if "(" in u else "")
                                     for u in rt.units if u.startswith("spectrum_fused")]}
            rt.destroy()
        result["value"] = result["cf32"]["zero_copy"]
        result["pcie_GBps"] = result["cf32"]["zero_copy_pcie_GBps"]
        return result

    dominant = "spectrum_fused" if not args.no_fuse else "spectrum.fft"
    algo_bytes = ALGO_BYTES_PER_SAMPLE * BATCHES * N_FFT

    def kernel_time(rt):
        # hipEvent pair around the kernel, in-graph, on the runtime's own stream.  Each event is a
        # packet of its own on the queue; an EMPTY pair recorded in the same graph (the kernel-less
        # "source" unit) measures two such packets back to back, live.  A pair that brackets a
        # kernel carries one packet's worth of that inside its interval, so half the empty-pair
        # time is subtracted; the result agrees with rocprofv3's per-dispatch average to ~4 %
        # (profiles/), the raw pair reads ~9 % high and the full subtraction ~12 % low.
        raw = rt.unit_mean_ms(dominant)
        pair = max(rt.event_overhead_ms(), 0.0)
        ms = raw - 0.5 * pair if raw > 0 else -1.0
        # a cycle-batched runtime's timed launches carry a whole ring period each: algorithmic bytes per LAUNCH scale with it
        kernel_time.cycles = max(rt.unit_mean_cycles(dominant), 1.0) if raw > 0 else 1.0
        return raw, pair, ms, (algo_bytes * kernel_time.cycles / (ms * 1e-3) / 1e9 if ms > 0 else None)

    def parity_check(rt, provider: str) -> dict:
        """The checker leg (never inside a timed region): the runtime that was just timed -- same graphs, same ring
        data -- runs one more ring period cycle by cycle (EVERY row of EVERY slot's range output is compared),
        then a whole-period graph replay plus a 3-cycle tail; the full spectrogram state the device ends with must
        equal, bit for bit, the oracle's replay of the same cycles from the state the device started this leg with."""
        t0 = time.perf_counter()
        from oracle import oracle  # the checker; only this leg and cpu_baseline() import it
        oracle.build()
        source, engine, spectrogram = rt._keep
        period, slots = max(rt.period, 1), args.slots
        rt.compute((-args.steps) % period, sync=True)  # every region starts at ring phase 0: back there
        rng = np.random.default_rng(rt._seed)
        data = [synth_slot(rng, s) for s in range(slots)]  # the generator that filled the ring, replayed
        state = spectrogram.state("frequencyBins").numpy().reshape(-1).copy()
        rows = np.arange(BATCHES)   # every row of every slot (rounds 1-5 sampled 64 of the 1024: VERDICT r05)
        refs, out_equal, max_err, bad_words = [], True, 0.0, 0
        for s in range(slots):
            rt.compute(1, sync=True)
            ref = oracle.chain_pass(data[s % len(data)], state, HEIGHT)  # also advances the oracle's state
            refs.append(ref)
            got = engine.buffer.numpy()[rows]
            same = np.array_equal(got.view(np.uint32), ref[rows].view(np.uint32))
            out_equal &= bool(same)
            if not same:
                bad_words += int(np.count_nonzero(got.view(np.uint32) != ref[rows].view(np.uint32)))
                max_err = max(max_err, float(np.max(np.abs(got.astype(np.float64) - ref[rows]))))
        tail_cycles = period + 3
        rt.compute(tail_cycles, sync=True)  # a whole-period graph replay and a span graph
        for c in range(tail_cycles):
            oracle.spectrogram(state, refs[c % slots], HEIGHT)
        ring_rows = 0
        if rt.batched:
            # cycle batching: the replay just ran the period (and the 3-cycle span) as ONE launch per unit, every cycle's
            # output in its ring slot -- the same row sample of EVERY slot must still equal the oracle's output
            for s in range(slots):
                got = engine.buffer.ring_select(s).numpy()[rows]
                same = np.array_equal(got.view(np.uint32), refs[s][rows].view(np.uint32))
                out_equal &= bool(same)
                ring_rows += int(rows.size)
                if not same:
                    bad_words += int(np.count_nonzero(got.view(np.uint32) != refs[s][rows].view(np.uint32)))
                    max_err = max(max_err, float(np.max(np.abs(got.astype(np.float64) - refs[s][rows]))))
            engine.buffer.ring_select((tail_cycles - 1) % slots)
        dev = spectrogram.state("frequencyBins").numpy().reshape(-1)
        state_equal = bool(np.array_equal(dev.view(np.uint32), state.view(np.uint32)))
        exact_provider = provider == "generic"
        return {"checked": True, "against": "oracle/jst_oracle.c chain pass (FFT restatement pinned to the reference's pocketfft)",
                "slots": slots, "rows_per_slot": int(rows.size), "output_rows": int(rows.size) * slots + ring_rows,
                "cycle_batched": bool(rt.batched), "batched_launch_output_rows": ring_rows,
                "cycles": slots + tail_cycles, "graph_replayed": bool(rt.graph_active),
                "output_bit_exact": bool(out_equal), "output_max_abs_err": max_err, "output_words_differing": bad_words,
                "spectrogram_state_bit_exact": state_equal, "spectrogram_state_words": int(state.size),
                # provider generic promises bits; provider fast promises exact bins (the state) and floats within 1e-5
                "bit_exact": bool(state_equal and (out_equal if exact_provider else max_err <= 1e-5)),
                "seconds": round(time.perf_counter() - t0, 2)}

    rt, elapsed = measure(args.provider)
    repeats_main, spread_main = measure.repeats, measure.spread
    percentiles_main = getattr(measure, "percentiles", [])

    # The path's optional exchange step (north_star: "optional RCCL-over-xGMI reduce"), OUTSIDE the timed region -- no
    # collective sits on the data path.  With RCCL as the control plane's backend every rank builds the LIBRARY's
    # communicator (csrc/jst/comm.cc: RCCL dlopen'ed behind the C ABI, no torch on the data) and times the two
    # collectives the path has: the U32[256, 4096] hit counts of the exact multi-GPU Spectrogram (4 MiB) and config 5's
    # averaged F32[65536] trace (256 KiB).  The line reports how many ranks RCCL saw.
    collective = None
    c5_multi = None
    if backend != "gloo" and (world > 1 or (js.comm_available() and not args.no_alt)):
        try:  # the measured line must not depend on the exchange step (a rank-symmetric failure skips it on every rank)
            import cyberether_amd.distributed as D
            # N = 1: a REAL one-rank RCCL communicator (jst_comm_init(0, 1, id)): the same dlopen, enum slice, stream ordering and
            # divide kernel as at N > 1, on this GPU -- so that the collective's own cost is a measured number at every N
            comm = D.library_comm() if world > 1 else js.Comm(0, 1, js.comm_unique_id())
            counts = js.Tensor.from_numpy(np.full((HEIGHT, N_FFT), rank + 1, np.uint32))
            trace = js.Tensor.from_numpy(np.full((65536,), float(rank + 1), np.float32))

            def timed_allreduce(t, average):
                for _ in range(3):
                    comm.all_reduce(t, "sum", average=average, stream=rt.stream)
                rt.synchronize()
                barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    comm.all_reduce(t, "sum", average=average, stream=rt.stream)
                rt.synchronize()
                return (time.perf_counter() - t0) / 20 * 1e6
            counts.copy_from(np.full((HEIGHT, N_FFT), rank + 1, np.uint32))
            comm.all_reduce(counts, "sum", stream=rt.stream)
            rt.synchronize()
            ok = bool(np.all(counts.numpy() == world * (world + 1) // 2))
            collective = {"library": "RCCL behind the C ABI (jst_comm_allreduce, csrc/jst/comm.cc)", "rccl_ranks": comm.world,
                          "rccl_ranks_match_world": bool(comm.world == world), "uses_rccl": comm.uses_rccl, "sum_of_counts_exact": ok,
                          "allreduce_us": {"u32_counts_4MiB": round(timed_allreduce(counts, False), 1),
                                           "f32_trace_256KiB_average": round(timed_allreduce(trace, True), 1)},
                          "in_timed_region": False}
            if comm.world != world or not comm.uses_rccl:
                print(f"[bench] WARNING: RCCL saw {comm.world} ranks (uses_rccl={comm.uses_rccl}) for a world of {world}", file=sys.stderr)
            if world > 1:
                # BASELINE configs[4] as it reads: one 65536-point spectrum stream per GPU (16 batches per cycle, Window -> FFT ->
                # Amplitude -> Range -> Lineplot average), the averaged PSD all-reduced over RCCL / xGMI once per reporting interval
                # -- INSIDE this timed loop, through the library's communicator on the runtime's stream (no torch on the data).
                n5, b5, interval, cycles5 = 65536, 16, 25, 200
                rng5 = np.random.default_rng(1240 + rank)   # seeds 1240..1247
                t5 = np.arange(n5) / 2.0e6
                x5 = (np.exp(2j * np.pi * (100.25 + rank) * 2.0e6 / n5 * t5)[None, :] +
                      1e-3 * (rng5.standard_normal((b5, n5)) + 1j * rng5.standard_normal((b5, n5)))).astype(np.complex64)
                src5 = js.Tensor.from_numpy(x5, batch=0, sample=1)
                eng5 = js.SpectrumEngine(src5, enable_scale=True, range_min=-100.0, range_max=0.0)
                lp5 = js.Module("lineplot", {"averaging": 8}, {"signal": eng5.buffer}, "psd")
                rt5 = js.Runtime(eng5.modules + [lp5], graph=True, fuse=True)
                own = lp5.state("averagingBuffer")
                merged = js.Tensor.create("hip", "F32", (n5,))   # the mean goes to a trace of its own: a rank's IIR state stays its own
                rt5.compute(interval, sync=True)
                barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done = 0
                while done < cycles5:
                    rt5.compute(interval, sync=False)
                    merged.copy_from_tensor(own, stream=rt5.stream)   # 256 KiB device copy behind the span, in front of the collective
                    comm.all_reduce(merged, "sum", average=True, stream=rt5.stream)
                    done += interval
                rt5.synchronize()
                torch.cuda.synchronize()
                dt5 = time.perf_counter() - t0
                t_max = torch.tensor([dt5], dtype=torch.float64, device="cuda")
                dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
                dt5 = float(t_max.item())
                # the merged trace must be the mean of the ranks' own traces
                mine = torch.from_numpy(own.numpy().astype(np.float64)).cuda()
                dist.all_reduce(mine, op=dist.ReduceOp.SUM)
                mean_ok = bool(np.allclose(merged.numpy(), (mine / world).cpu().numpy(), rtol=0, atol=1e-6))
                c5_multi = {"config": f"configs[4]: {world} independent 65536-point spectrum streams (one per GPU), PSD all-reduce "
                                      f"(RCCL, library communicator) every {interval} cycles inside the timed loop",
                            "n_gpus": world, "value": world * b5 * n5 * cycles5 / dt5 / 1e6, "unit": "MS/s",
                            "us_per_cycle": dt5 / cycles5 * 1e6, "merged_trace_is_mean_of_rank_traces": mean_ok,
                            "roofline_frac_per_gpu": 28.0 * b5 * n5 / (dt5 / cycles5) / 8e12}
                rt5.destroy()
        except Exception as exc:
            print(f"[bench] collective leg failed: {exc!r}", file=sys.stderr)
            collective = collective or {"error": repr(exc)}
    samples = float(args.steps) * BATCHES * N_FFT * world
    kernel_ms_raw, pair_ms, kernel_ms, achieved = kernel_time(rt)
    cycles_per_launch = kernel_time.cycles

    line = None
    if rank == 0:
        traffic, traffic_src, rocprof_us = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_summary.py from --pmc passes
        if os.path.exists(pmc):
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_hash import kernel_sources_sha256
            doc = json.load(open(pmc))
            # one record per (provider, launch form): "<provider>@<cycles per launch>", the bare provider key = the default form
            rec = doc.get(f"{args.provider}@{int(round(cycles_per_launch))}", doc.get(args.provider, {}))
            if rec.get("kernel_sources_sha256") == kernel_sources_sha256() and \
                    int(rec.get("cycles_per_launch", 1)) == int(round(cycles_per_launch)):
                traffic = rec.get("spectrum_fused_hbm_bytes_per_launch")
                rocprof_us = rec.get("rocprofv3_kernel_us_mean")
                traffic_src = {"source": rec.get("source"), "kernel": rec.get("kernel"),
                               "kernel_sources_sha256": rec.get("kernel_sources_sha256"),
                               "cycles_per_launch": rec.get("cycles_per_launch", 1),
                               "fetch_size_kib_mean": rec.get("fetch_size_kib_mean"),
                               "write_size_kib_mean": rec.get("write_size_kib_mean"), "correction": rec.get("correction")}
            else:
                traffic_src = {"source": None, "stale": "profiles/pmc_traffic.json was measured on other kernel sources "
                                                         "(another provider, or another launch form): not quoted"}
        step_ms = elapsed / args.steps * 1e3
        step_bytes = STEP_BYTES_PER_SAMPLE * BATCHES * N_FFT
        frac_rocprof = (algo_bytes * cycles_per_launch / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None
        line = {
            "metric": baseline_metric(),
            "value": samples / elapsed / 1e6,
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: Window->4096-pt FFT->Amplitude->Range->Spectrogram(h=256), "
                                   "1024 batches cf32 per step, hipGraph capture",
                       "batches": BATCHES, "fft_size": N_FFT, "ring_slots": args.slots,
                       "graph": rt.graph_active, "fused": not args.no_fuse,
                       "provider": args.provider,
                       "provider_contract": ("integer bin work bit-exact (proven per input power), float spectra within 4e-7 "
                                             "absolute of the reference CPU path (north_star: 1e-5)" if args.provider == "fast"
                                             else "every output bit-identical to the reference CPU path"),
                       "pipelined": args.pipeline, "combined": args.combine,
                       # cycle batching (JST_RUNTIME_BATCH): the resident ring slots of a period run as ONE launch per
                       # unit; every step is still one pass over one CF32[1024, 4096] batch (its output in its ring slot,
                       # its own decay + hit update of the Spectrogram state); alt_per_cycle_launch is the other form
                       "cycle_batching": bool(rt.batched), "cycles_per_launch": cycles_per_launch,
                       "untimed_init_steps": 2 * max(rt.period, 1) + (-args.warmup) % max(rt.period, 1)
                                             + args.steps + (-args.steps) % max(rt.period, 1),
                       "repeats": repeats_main, "region_ms_min_max": [round(spread_main[0] * 1e3, 4),
                                                                      round(spread_main[1] * 1e3, 4)],
                       "region_ms_p10_p50_p90": [round(v * 1e3, 4) for v in percentiles_main],  # `value` is the MEAN over the regions
                       "ring_period": rt.period, "backend": backend if world > 1 else None,
                       # what the Spectrogram reads: the fused kernel's one-byte row indices (its side output) or the values
                       "spectrogram_input": "row indices (U8 side output of the fused kernel)"
                                            if any(u.endswith("+indices") for u in rt.units) else "values (F32)",
                       "units_ms": {u.split("(")[0]: rt.unit_mean_ms(u) for u in rt.units
                                    if rt.unit_mean_ms(u) > 0},
                       "sharding": "independent batches per GPU, no data-path collective",
                       "collective": collective},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         # `frac` is the CONSERVATIVE figure: the tracked rocprofv3 summary's mean dispatch duration of this kernel
                         # (profiles/, same kernel sources by hash, same launch form; the profiler costs the kernel a few percent)
                         # when there is one, else the event pair's; both are always stated beside it
                         "frac": frac_rocprof if frac_rocprof else ((achieved / HBM_PEAK_GBS) if achieved else None),
                         "frac_source": "rocprofv3 (profiles/)" if frac_rocprof else "hipEvent pair (no matching rocprofv3 record)",
                         "frac_event_pair": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "frac_rocprof": frac_rocprof,
                         "rocprofv3_kernel_us": rocprof_us,
                         "traffic": traffic, "traffic_provenance": traffic_src, "kernel_ms": kernel_ms,
                         "kernel_ms_method": "hipEvent pair on the runtime's stream around the kernel's eager launches "
                                             "inside the timed region (every 16th ring period: one cycle of it, or -- cycle "
                                             "batching -- the whole period as one launch per unit), minus half of an empty "
                                             "pair measured the same way; rocprofv3's per-dispatch mean (profiles/) is the "
                                             "cross-check, not this number's source",
                         "kernel_ms_event_pair_raw": kernel_ms_raw,
                         "event_pair_overhead_ms": pair_ms,
                         "algorithmic_bytes_per_launch": algo_bytes * cycles_per_launch,
                         "cycles_per_launch": cycles_per_launch,
                         "transforms_per_launch": BATCHES * cycles_per_launch,
                         # what the launch moves beyond that: +1 B/sample written when the kernel also emits the
                         # Spectrogram's row indices (the consumer then reads 1 B/sample instead of 4)
                         "side_output_bytes_per_launch": float(BATCHES * N_FFT) * cycles_per_launch if any(u.endswith("+indices") for u in rt.units) else 0.0,
                         # the whole step on SURVEY 8(d)'s 14 B/sample (spectrum 12 + spectrogram state 2), per rank
                         "step_bytes": step_bytes,
                         "step_achieved": step_bytes / (step_ms * 1e-3) / 1e9,
                         "step_frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if not args.no_parity and not args.no_fuse and not args.pipeline:
            try:
                line["parity"] = parity_check(rt, args.provider)
            except Exception as exc:  # a missing compiler for the checker must not lose the measurement
                line["parity"] = {"checked": False, "error": repr(exc)}
        else:
            line["parity"] = {"checked": False, "reason": "--no-parity / unfused / pipelined run"}
        if world == 1 and not args.no_fuse and not args.no_alt:
            # second measurement: the same chain with the OTHER amplitude/range provider ("generic" = every float
            # bit-identical to the reference CPU path; "fast" = bins exact, floats within 4e-7), with its own roofline
            # figure and parity stamp
            other = "generic" if args.provider == "fast" else "fast"
            rt2, elapsed2 = measure(other, seed_offset=0)
            raw2, pair2, ms2, ach2 = kernel_time(rt2)
            line["alt_provider"] = {"provider": other, "value": samples / elapsed2 / 1e6, "unit": "MS/s",
                                    "ms_per_step": elapsed2 / args.steps * 1e3, "kernel_ms": ms2,
                                    "roofline_frac": (ach2 / HBM_PEAK_GBS) if ach2 else None,
                                    "step_frac": step_bytes / (elapsed2 / args.steps) / 1e9 / HBM_PEAK_GBS}
            if not args.no_parity:
                try:
                    line["alt_provider"]["parity"] = parity_check(rt2, other)
                except Exception as exc:
                    line["alt_provider"]["parity"] = {"checked": False, "error": repr(exc)}
            rt2.destroy()
        if world == 1 and rt.batched and not args.no_alt and not args.no_graph:
            # The decision JST_RUNTIME_EAGER_SPANS asks for, on this run's own evidence: a cycle-batched span is two kernel
            # launches; replayed as a two-node hipGraph (the default: configs[1] names hipGraph capture) or submitted directly.
            js.debug_set("JST_RUNTIME_EAGER_SPANS", "1")
            try:
                rt6, elapsed6 = measure(args.provider, seed_offset=0)
            finally:
                js.debug_set("JST_RUNTIME_EAGER_SPANS", None)
            line["alt_eager_spans"] = {"value": samples / elapsed6 / 1e6, "unit": "MS/s", "ms_per_step": elapsed6 / args.steps * 1e3,
                                       "step_frac": step_bytes / (elapsed6 / args.steps) / 1e9 / HBM_PEAK_GBS,
                                       "what": "the same cycle-batched spans submitted as direct kernel launches instead of a "
                                               "replayed hipGraph of two kernel nodes; the headline keeps the graph"}
            rt6.destroy()
        if world == 1 and rt.batched and not args.no_alt:
            # the same chain with ONE LAUNCH PER UNIT AND CYCLE (rounds 1-2's form; --no-batch makes it the headline):
            # its own kernel time, roofline fraction and parity stamp
            rt5, elapsed5 = measure(args.provider, seed_offset=0, batch=False)
            raw5, pair5, ms5, ach5 = kernel_time(rt5)
            line["alt_per_cycle_launch"] = {"value": samples / elapsed5 / 1e6, "unit": "MS/s",
                                            "ms_per_step": elapsed5 / args.steps * 1e3, "kernel_ms": ms5,
                                            "cycles_per_launch": kernel_time.cycles,
                                            "roofline_frac": (ach5 / HBM_PEAK_GBS) if ach5 else None,
                                            "step_frac": step_bytes / (elapsed5 / args.steps) / 1e9 / HBM_PEAK_GBS,
                                            "units_ms": {u.split("(")[0]: rt5.unit_mean_ms(u) for u in rt5.units
                                                         if rt5.unit_mean_ms(u) > 0}}
            if not args.no_parity:
                try:
                    line["alt_per_cycle_launch"]["parity"] = parity_check(rt5, args.provider)
                except Exception as exc:
                    line["alt_per_cycle_launch"]["parity"] = {"checked": False, "error": repr(exc)}
            rt5.destroy()
        if world == 1 and not args.no_fuse and not args.no_alt and not args.pipeline and not args.combine:
            # the round-1 definition of the headline (provider generic, one launch per unit and cycle), so that a reader can
            # follow r01 -> r04 on ONE definition
            rt6, elapsed6 = measure("generic", seed_offset=0, batch=False)
            line["value_generic_per_cycle"] = {"value": samples / elapsed6 / 1e6, "unit": "MS/s",
                                               "ms_per_step": elapsed6 / args.steps * 1e3,
                                               "what": "provider generic (every float bit-identical to the reference CPU "
                                                       "path), one launch per unit and compute cycle: rounds 1-2's headline form"}
            rt6.destroy()
        if world == 1 and not args.pipeline and not args.combine and not args.no_graph and not args.no_fuse and not args.no_alt:
            # informational third measurement: one kernel per cycle (spectrum of cycle k + spectrogram of cycle k - 1);
            # kernel_ms is then the combined kernel's and is not comparable with the 12 B/sample roofline above
            rt4, elapsed4 = measure(args.provider, seed_offset=0, pipeline=False, combine=True, batch=False)
            line["alt_combined"] = {"value": samples / elapsed4 / 1e6, "unit": "MS/s",
                                    "ms_per_step": elapsed4 / args.steps * 1e3,
                                    "units": [u.split("(")[0] for u in rt4.units if not u.startswith("spectrum.")]}
            rt4.destroy()
        if world == 1 and not args.no_host_fed and not args.no_alt:
            try:
                line["host_fed"] = host_fed(args.provider)
            except Exception as exc:  # the headline must not depend on it
                line["host_fed"] = {"error": repr(exc)}

    rt.destroy()
    if rank == 0 and world == 1 and not args.no_alt and not args.no_configs:
        line["configs"] = other_configs(js)
    if rank == 0 and world == 1 and not args.no_alt:
        line["reference_driven"] = reference_driven(args.slots, args.steps, args.provider)
    if rank == 0 and c5_multi is not None:
        line["configs"] = [c5_multi]
    if world > 1:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        # after the process group is gone (N > 1: the other ranks have left), so a SCALE line carries it too
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
