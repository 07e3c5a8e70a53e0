"""CPU baseline of the hot path for bench.py (test/bench infrastructure): the spectrum chain
Window(x)Invert -> Multiply -> FFT -> Amplitude -> Range -> Spectrogram in dense C loops
(oracle/jst_oracle.c: jst_oracle_chain_bench), the FFT through the reference's own pocketfft
(oracle/_ref) when that library is present, timed nanobench-style like src/benchmark.cc:100-106,175-186
(warm-up, epochs of >= 100 ms, median).  Run as a module it prints one JSON line -- bench.py starts one
process per host core for the `all_cores` replica figure (the reference's compute path is single-threaded,
fft/module_impl_native_cpu.cc:1-2, so independent replicas over disjoint batch shards are the fair way to
use every core)."""
from __future__ import annotations

import argparse
import ctypes as C
import json

import numpy as np

from oracle import oracle

N_FFT, HEIGHT = 4096, 256


def run(rows: int, epoch_s: float, epochs: int, use_ref: bool = True, seed: int = 4321) -> dict:
    lib = oracle.lib()
    fft_ptr, kind = None, "port"
    if use_ref and oracle.have_ref():
        fft_ptr = C.cast(oracle.ref().ref_fft_c2c, C.c_void_p)
        kind = "reference"
    rng = np.random.default_rng(seed)
    n = np.arange(N_FFT, dtype=np.float64)
    bins_ = (100.25 + np.arange(rows, dtype=np.float64)) % N_FFT
    phase = 2.0 * np.pi * bins_[:, None] * n[None, :] / N_FFT
    x = np.empty((rows, N_FFT), np.complex64)
    x.real = np.cos(phase) + rng.standard_normal((rows, N_FFT)).astype(np.float32) * np.float32(1e-3)
    x.imag = np.sin(phase) + rng.standard_normal((rows, N_FFT)).astype(np.float32) * np.float32(1e-3)
    window = oracle.invert(oracle.window(N_FFT))
    coeff = oracle.amplitude_coeff(N_FFT)
    scale, offset = oracle.range_coeffs(-100.0, 0.0)
    product = np.empty((rows, N_FFT), np.complex64)
    spectrum = np.empty((rows, N_FFT), np.complex64)
    out = np.empty((rows, N_FFT), np.float32)
    state = np.zeros(N_FFT * HEIGHT, np.float32)
    rates = (C.c_double * epochs)()
    f32p = C.POINTER(C.c_float)
    fn = lib.jst_oracle_chain_bench
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, f32p, f32p, C.c_uint64, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_uint64,
                   f32p, f32p, f32p, f32p, C.c_double, C.c_uint32, C.POINTER(C.c_double)]
    p = lambda a: a.ctypes.data_as(f32p)
    median = fn(fft_ptr, p(x), p(window), rows, N_FFT, coeff, scale, offset, HEIGHT, p(product), p(spectrum),
                p(out), p(state), epoch_s, epochs, rates)
    return {"samples_per_s": float(median), "kind": kind, "rows": rows, "epochs": epochs, "epoch_s": epoch_s,
            "epoch_rates_MSps": [round(r / 1e6, 2) for r in rates]}


def run_reference_runtime(rows: int, epoch_s: float, epochs: int, spectrogram: bool = True, seed: int = 4321) -> dict:
    """The SAME chain through the REFERENCE'S OWN RUNTIME: oracle/_ref/libref_jetstream.so is the reference's core, scheduler,
    native-CPU runtime, modules and blocks compiled in place (oracle/ref_jetstream_build.sh).  A Flowgraph holds a source,
    the `spectrum_engine` block (cast -> window -> invert -> reshape -> multiply -> fft -> amplitude -> range: its own
    module_impl_native_cpu.cc files, AutomaticIterator and all) and the `spectrogram` block; Flowgraph::compute() is timed
    nanobench-style in a C loop (src/benchmark.cc:100-106,175-186 shape).  Returns None when the library is absent."""
    from oracle import ref_jetstream as rj
    if not rj.available():
        return None
    rng = np.random.default_rng(seed)
    n = np.arange(N_FFT, dtype=np.float64)
    bins_ = (100.25 + np.arange(rows, dtype=np.float64)) % N_FFT
    phase = 2.0 * np.pi * bins_[:, None] * n[None, :] / N_FFT
    x = np.empty((rows, N_FFT), np.complex64)
    x.real = np.cos(phase) + rng.standard_normal((rows, N_FFT)).astype(np.float32) * np.float32(1e-3)
    x.imag = np.sin(phase) + rng.standard_normal((rows, N_FFT)).astype(np.float32) * np.float32(1e-3)
    with rj.RefFlowgraph() as fg:
        fg.source("src", x, sample=1, batch=0)
        assert fg.block("eng", "spectrum_engine", {"enableScale": True, "rangeMin": -100.0, "rangeMax": 0.0},
                        {"buffer": "src:signal"}) == 0 and fg.state("eng") == 2
        if spectrogram:
            assert fg.block("spec", "spectrogram", {"height": HEIGHT}, {"signal": "eng:buffer"}) == 0 and fg.state("spec") == 2
        rates = fg.compute_timed(epoch_s, epochs)
    per_s = sorted(rates)[len(rates) // 2] * rows * N_FFT
    return {"samples_per_s": float(per_s), "kind": "reference-runtime", "rows": rows, "epochs": epochs, "epoch_s": epoch_s,
            "spectrogram": spectrogram, "epoch_rates_MSps": [round(r * rows * N_FFT / 1e6, 2) for r in rates]}


def run_configs0(epoch_s: float = 0.1, epochs: int = 11, use_ref: bool = True) -> dict:
    """BASELINE configs[0], like for like: ONE batch of 4096 cf32 (an off-bin CW tone) through
    FFT -> Amplitude, one op per compute(), nanobench-style (epochs >= 100 ms, median) as src/benchmark.cc:100-106,
    175-186 does; the FFT is the reference's own pocketfft when oracle/_ref is present."""
    lib = oracle.lib()
    fft_ptr, kind = None, "port"
    if use_ref and oracle.have_ref():
        fft_ptr = C.cast(oracle.ref().ref_fft_c2c, C.c_void_p)
        kind = "reference"
    x, _ = oracle.signal_cosine(N_FFT, 1.0, 100.25 * 2.0e6 / N_FFT, 2.0e6)
    x = np.ascontiguousarray(x)
    spectrum = np.empty(N_FFT, np.complex64)
    out = np.empty(N_FFT, np.float32)
    per_op = (C.c_double * epochs)()
    f32p = C.POINTER(C.c_float)
    fn = lib.jst_oracle_fft_amplitude_bench
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_float, f32p, f32p, C.c_double, C.c_uint32, C.POINTER(C.c_double)]
    p = lambda a: a.ctypes.data_as(f32p)
    median = fn(fft_ptr, p(x.view(np.float32)), N_FFT, oracle.amplitude_coeff(N_FFT), p(spectrum.view(np.float32)),
                p(out), epoch_s, epochs, per_op)
    return {"us_per_op": median * 1e6, "samples_per_s": N_FFT / median, "kind": kind, "epochs": epochs,
            "epoch_s": epoch_s, "epoch_us_per_op": [round(v * 1e6, 3) for v in per_op]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--epoch-s", type=float, default=0.2)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--seed", type=int, default=4321)
    ap.add_argument("--reference-runtime", action="store_true", help="time the chain through the compiled reference's own runtime")
    a = ap.parse_args()
    print(json.dumps(run(a.rows, a.epoch_s, a.epochs, not a.no_ref, a.seed)), flush=True)
