"""Where the fixed cost of a timed region goes (bench.py's region(): compute(K) + runtime sync + torch.cuda.synchronize):
wall time per region for K = 20 under four bracketing forms, 300 regions each.  Diagnostic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cyberether_amd import jetstream as js

B, N, H, SLOTS, K = 1024, 4096, 256, 16, 20
src = js.Module("ring_source", {"batches": B, "samples": N, "slots": SLOTS}, {}, "source")
buf = src.output("buffer")
rng = np.random.default_rng(1)
for s in range(SLOTS):
    buf.ring_select(s).copy_from((rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))).astype(np.complex64) * np.float32(0.05))
buf.ring_select(0)
eng = js.SpectrumEngine(buf, enable_scale=True, range_min=-100.0, range_max=0.0, provider="fast")
spec = js.Module("spectrogram", {"height": H}, {"signal": eng.buffer}, "spectrogram")
rt = js.Runtime([src] + eng.modules + [spec], graph=True, fuse=True)
rt.compute(2 * SLOTS, sync=True); rt.compute(K, sync=True); rt.compute((-K) % SLOTS, sync=True)

def run(form, reps=300):
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if form == "compute+sync+torch":
            rt.compute(K, sync=False); rt.synchronize(); torch.cuda.synchronize()
        elif form == "compute(sync)+torch":
            rt.compute(K, sync=True); torch.cuda.synchronize()
        elif form == "compute(sync)":
            rt.compute(K, sync=True)
        elif form == "compute+torch":
            rt.compute(K, sync=False); torch.cuda.synchronize()
        tot += time.perf_counter() - t0
        rt.compute((-K) % SLOTS, sync=True)
    return tot / reps
for form in ("compute+sync+torch", "compute(sync)+torch", "compute(sync)", "compute+torch", "compute+sync+torch"):
    dt = run(form)
    print(f"{form:22s} {dt * 1e6:8.2f} us per region = {dt * 1e6 / K:6.3f} us per step")
