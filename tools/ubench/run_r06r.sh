# constant plans for 16384 / 32768 / 131072 / 262144 points against the run-time-plan kernels: Window -> FFT -> Amplitude -> Range, us per cycle
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_tiled_persistent.py tests/test_gpu_fft.py -q -m gpu -x 2>&1 | tail -3
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
for rep in 1 2; do for v in base tiled_prev; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  python - <<PY
import sys, os, time
import numpy as np
sys.path.insert(0, "$ROOT")
import torch
import cyberether_amd.jetstream as js
out = []
for provider in ("fast", "generic"):
    for n, b in ((16384, 512), (16384, 32), (32768, 256), (32768, 16), (131072, 64), (131072, 8), (262144, 32), (262144, 4)):
        rng = np.random.default_rng(1)
        x = (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))).astype(np.complex64)
        src = js.Tensor.from_numpy(x, batch=0, sample=1)
        eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0, provider=provider)
        rt = js.Runtime(eng.modules, graph=True, fuse=True)
        rt.compute(10, sync=True); torch.cuda.synchronize(); t0 = time.perf_counter(); rt.compute(100, sync=False); rt.synchronize(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        out.append(f"{provider[0]} {b}x{n}: {dt*1e6:.1f} us ({28.0*b*n/dt/8e12:.3f})")
        rt.destroy()
print("$v", " | ".join(out))
PY
done; done
cp /tmp/base.so $L
