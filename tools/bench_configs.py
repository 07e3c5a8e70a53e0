#!/usr/bin/env python3
"""Secondary benchmark lines for the other BASELINE.json configs (1 GPU, device-resident inputs):
  C1  signal_generator -> Window -> 4096-pt FFT -> Amplitude, 1 batch (latency of one cycle)
  C3  Filter block: 251-tap band-pass + /10 resampling on CF32[100, 159750] (16 MS per cycle)
  C4  WBFM chain: Filter(20 MS/s -> 200 kS/s) -> FM(wide) -> Decimator(/4)
  C5  65536-pt Window -> FFT -> Amplitude -> Range -> Lineplot average, 16 batches (one stream)
Prints one JSON line per config (informational; bench.py is the contract benchmark)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


def roofline(bytes_per_sample, samples_per_cycle, seconds_per_cycle, what):
    """Whole-chain HBM roofline of a config: SURVEY 8(d)'s ALGORITHMIC bytes per input sample x the samples one cycle
    processes / the measured cycle time, against the 8 TB/s spec peak."""
    achieved = bytes_per_sample * samples_per_cycle / seconds_per_cycle / 1e9
    return {"bound": "hbm", "bytes_per_sample": bytes_per_sample, "what": what, "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}


def timed(rt, cycles, warmup):
    import torch
    rt.compute(warmup, sync=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rt.compute(cycles, sync=False)
    rt.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / cycles


def main():
    import torch
    torch.cuda.set_device(0)
    import cyberether_amd.jetstream as js
    rng = np.random.default_rng(1235)
    only = set(a.upper() for a in sys.argv[1:])  # e.g. "C3 C5": run just these (for rocprofv3)
    want = lambda tag: not only or tag in only
    class _Out(list):
        def append(self, line):
            print(json.dumps(line), flush=True)
    out = _Out()

    # ---- C1 ------------------------------------------------------------------------------------
    n, fs = 4096, 2.0e6
    if want('C1'):
        run_c1(js, out, n, fs)
    if want('C3'):
        run_c3(js, out, rng)
    if want('C4'):
        run_c4(js, out)
    if want('C5'):
        run_c5(js, out, rng)
    if want('AM'):
        run_am(js, out)


def run_c1(js, out, n, fs):
    gen = js.Module("signal_generator", {"signalType": "cosine", "signalDataType": "CF32",
                                         "sampleRate": fs, "frequency": 100.25 * fs / n,
                                         "bufferSize": n}, {}, "cw")
    eng = js.SpectrumEngine(gen.output("signal"), enable_scale=False)
    rt = js.Runtime([gen] + eng.modules, graph=True, fuse=True)
    dt = timed(rt, 200, 20)
    out.append({"config": "C1: CW tone -> Window -> 4096-pt FFT -> Amplitude, 1 batch", "us_per_cycle": dt * 1e6,
                "MS_per_s": n / dt / 1e6, "note": "latency bound (one transform); includes the serial tone generator"})
    rt.destroy()
    src1 = js.Module("ring_source", {"batches": 1, "samples": n, "slots": 4}, {}, "iq")
    eng = js.SpectrumEngine(src1.output("buffer"), enable_scale=False)
    rt = js.Runtime([src1] + eng.modules, graph=True, fuse=True)
    dt = timed(rt, 400, 40)
    out.append({"config": "C1b: resident IQ -> Window -> 4096-pt FFT -> Amplitude, 1 batch (no generator)",
                "us_per_cycle": dt * 1e6, "MS_per_s": n / dt / 1e6,
                "note": "one fused launch per cycle inside a captured graph: launch-latency bound"})
    rt.destroy()



def run_c3(js, out, rng):
    b, s, taps, sr, bw = 100, 159750, 251, 20e6, 2e6
    t = np.arange(b * s) / sr
    x = (np.exp(2j * np.pi * 0.3e6 * t) + np.exp(2j * np.pi * 4.0e6 * t)).astype(np.complex64)
    x += (0.01 * (rng.standard_normal(b * s) + 1j * rng.standard_normal(b * s))).astype(np.complex64)
    src = js.Tensor.from_numpy(x.reshape(b, s), batch=0, sample=1)
    blk = js.Filter(src, sr, bw, [0.0], taps, 1)
    rt = js.Runtime(blk.modules, graph=True, fuse=True)
    dt = timed(rt, 20, 3)
    out.append({"config": "C3: Filter block 251 taps, /10, CF32[100,159750] (conv 160000 = 8*8*4*5^4)",
                "ms_per_cycle": dt * 1e3, "MS_per_s_in": b * s / dt / 1e6, "plan": blk.plan,
                "units": rt.units, "note": "FFT overlap-add: tiled 160000-pt FFT with the pad fused in, fold of the never-materialised product",
                "roofline": roofline(32.0, b * s, dt, "FFT overlap-add view: forward transform 8 r + 8 w + 8 r, product/fold/inverse/overlap 8 (SURVEY 8d)")})
    exact = blk.buffer.numpy()
    rt.destroy()
    blk = js.Filter(src, sr, bw, [0.0], taps, 1, provider="fast")
    rt = js.Runtime(blk.modules, graph=True, fuse=True)
    dt = timed(rt, 50, 5)
    # same input every cycle: after the first cycle the history equals the chain's overlap state
    fast = blk.buffer.numpy()
    err = float(np.max(np.abs(fast[1:] - exact[1:])) / np.max(np.abs(exact)))
    out.append({"config": "C3-fast: Filter block, provider fast = one direct-form polyphase FIR + /10 kernel",
                "ms_per_cycle": dt * 1e3, "MS_per_s_in": b * s / dt / 1e6, "units": rt.units,
                "max_err_vs_fft_chain_rel_peak": err,
                "algorithmic_GBps": (b * s * 8 + b * s // 10 * 8) / dt / 1e9,
                "roofline": roofline(8.8, b * s, dt, "time-domain view: 8 B read + 0.8 B written per input sample (SURVEY 8d)")})
    rt.destroy()



def run_c4(js, out):
    b, s, taps, sr, bw = 10, 202400, 101, 20e6, 200e3  # conv 202500 = 2^2*3^4*5^4
    tt = np.arange(b * s) / sr
    audio = np.sin(2 * np.pi * 1e3 * tt) * 0.45 + 0.1 * np.sin(2 * np.pi * 19e3 * tt)
    x = np.exp(2j * np.pi * 75e3 * np.cumsum(audio) / sr).astype(np.complex64)
    src = js.Tensor.from_numpy(x.reshape(b, s), batch=0, sample=1)
    filt = js.Filter(src, sr, bw, [0.0], taps, 1)
    squeeze = js.Module("squeeze_dims", {"axis": 1}, {"buffer": filt.buffer}, "squeeze_head")
    iq = squeeze.output("buffer").set_axes(batch=0, sample=1)
    fm = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3}, {"signal": iq}, "fm")
    dec = js.Decimator(fm.output("signal"), 4)
    rt = js.Runtime(filt.modules + [squeeze, fm] + dec.modules, graph=True, fuse=True)
    dt = timed(rt, 20, 3)
    out.append({"config": "C4: WBFM 20 MS/s -> Filter(/100) -> FM wide 75us -> Decimator(/4)",
                "ms_per_cycle": dt * 1e3, "MS_per_s_in": b * s / dt / 1e6,
                "audio_shape": list(dec.buffer.shape),
                "roofline": roofline(8.0 + 0.08 * 32.0, b * s, dt, "channel filter 8 B read per input sample + the /100 chain; the FM "
                                                                     "decode is latency bound (serial recurrences), not HBM bound"),
                "note": "FM stereo decode = serial recurrences per lane (1 lane here), run as software-pipelined wavefront stages with DPP lane pipelines inside (fm_wide_kernel); 17.9 ms with the one-thread walk (JST_FM_SERIAL=1)"})
    rt.destroy()
    # the same decoder on many stations at once: lanes are independent workgroups
    lanes, nb, ns = 64, 10, 2024
    tt = np.arange(nb * ns) / 200e3
    base = np.exp(2j * np.pi * 75e3 * np.cumsum(0.45 * np.sin(2 * np.pi * 1e3 * tt)) / 200e3)
    xs = np.stack([np.roll(base, 37 * l) for l in range(lanes)], axis=0).astype(np.complex64)   # [lanes, n]
    xs = np.ascontiguousarray(xs.reshape(lanes, nb, ns).transpose(1, 0, 2))                      # [nb, lanes, ns]
    t = js.Tensor.from_numpy(xs, batch=0, sample=2)
    fm = js.Module("fm", {"mode": "wide", "deemphasis": "75us", "sampleRate": 200e3}, {"signal": t}, "fm64")
    rt = js.Runtime([fm], graph=True)
    dt = timed(rt, 20, 3)
    out.append({"config": "C4b: FM wide 75us decode of 64 stations x 20240 samples at 200 kS/s (fm module only)",
                "ms_per_cycle": dt * 1e3, "stations_x_realtime": lanes * (nb * ns / 200e3) / dt,
                "note": "one workgroup per station: the decode time of one station covers all of them"})
    rt.destroy()



def run_am(js, out):
    """AM broadcast side chain with C4's shape: 20 MS/s -> Filter(/100) -> AM envelope + DC blocker -> Decimator(/4)."""
    b, s, taps, sr, bw = 10, 202400, 101, 20e6, 200e3
    tt = np.arange(b * s) / sr
    x = ((1.0 + 0.5 * np.cos(2 * np.pi * 1e3 * tt)) * np.exp(2j * np.pi * 10e3 * tt)).astype(np.complex64)
    src = js.Tensor.from_numpy(x.reshape(b, s), batch=0, sample=1)
    filt = js.Filter(src, sr, bw, [0.0], taps, 1)
    squeeze = js.Module("squeeze_dims", {"axis": 1}, {"buffer": filt.buffer}, "squeeze_head")
    iq = squeeze.output("buffer").set_axes(batch=0, sample=1)
    am = js.Module("am", {"sampleRate": 200e3, "dcAlpha": 0.995}, {"signal": iq}, "am")
    dec = js.Decimator(am.output("signal"), 4)
    rt = js.Runtime(filt.modules + [squeeze, am] + dec.modules, graph=True, fuse=True, timing=True)
    dt = timed(rt, 20, 3)
    out.append({"config": "AM: 20 MS/s -> Filter(/100) -> AM (dcAlpha 0.995) -> Decimator(/4)",
                "ms_per_cycle": dt * 1e3, "MS_per_s_in": b * s / dt / 1e6, "am_kernel_ms": rt.unit_mean_ms("am"),
                "note": "one lane: envelope and first difference in parallel, the DC-blocker recurrence walked by one thread out of LDS"})
    rt.destroy()
    lanes, n = 64, 1 << 18
    t = js.Tensor.from_numpy(np.ascontiguousarray(np.broadcast_to(x[:n], (lanes, n))), channel=0, sample=1)
    am = js.Module("am", {"sampleRate": 200e3}, {"signal": t}, "am64")
    rt = js.Runtime([am], graph=True)
    dt = timed(rt, 20, 3)
    out.append({"config": "AMb: AM on 64 lanes x 262144 samples (am module only)", "ms_per_cycle": dt * 1e3,
                "MS_per_s_per_lane": n / dt / 1e6, "MS_per_s_total": lanes * n / dt / 1e6})
    rt.destroy()


def run_c5(js, out, rng):
    n, b = 65536, 16
    x = (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True)
    lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
    rt = js.Runtime(eng.modules + [lp], graph=True, fuse=True)
    dt = timed(rt, 50, 5)
    out.append({"config": "C5 (per GPU): Window -> 65536-pt FFT -> Amplitude -> Range -> Lineplot avg, 16 batches",
                "us_per_cycle": dt * 1e6, "MS_per_s": b * n / dt / 1e6, "units": rt.units,
                "note": "65536-pt FFT = two LDS-tiled kernels (columns + blocks) with window / amplitude / range fused in; PSD "
                        "all-reduce is one 256 KiB RCCL all-reduce per reporting interval (cyberether_amd/distributed.py)",
                "roofline": roofline(28.0, b * n, dt, "two HBM passes: 8 r + 8 w + 8 r + 4 w per sample (SURVEY 8d); 1 Mi samples "
                                                     "per cycle: launch/occupancy bound")})
    rt.destroy()
    # the same chain on a resident ring of 16 slots (16 x 8 MiB), per cycle and CYCLE-BATCHED (runs of consecutive slots as
    # one columns / blocks launch pair, the lineplot riding behind as a sink): what three launches per 1 Mi samples cost
    for batch in (False, True):
        slots = 16
        ring = js.Module("ring_source", {"batches": b, "samples": n, "slots": slots}, {}, "iq")
        buf = ring.output("buffer")
        for s in range(slots):
            buf.ring_select(s).copy_from(np.roll(x, s, axis=0))
        buf.ring_select(0)
        eng = js.SpectrumEngine(buf, enable_scale=True)
        lp = js.Module("lineplot", {"averaging": 8}, {"signal": eng.buffer}, "psd")
        rt = js.Runtime([ring] + eng.modules + [lp], graph=True, fuse=True, batch=batch)
        dt = timed(rt, 320, 48)
        out.append({"config": "C5 on a resident ring of 16 slots" + (", cycle-batched" if batch else ", one launch per unit and cycle"),
                    "us_per_cycle": dt * 1e6, "MS_per_s": b * n / dt / 1e6, "batched": bool(rt.batched),
                    "roofline": roofline(28.0, b * n, dt, "as C5")})
        rt.destroy()



if __name__ == "__main__":
    main()
