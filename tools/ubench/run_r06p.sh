# register budget of the tiled kernels (launch bound: 6 / 5 / 4 wavefronts per SIMD = 80 / 96 / 128 VGPRs): the run-time-plan kernels spill at 80
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
for rep in 1 2; do for v in base tiled_w5 tiled_w4; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  python tools/bench_multi_fm.py 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v multi-fm', round(d['us_per_cycle'],2))"
  python - <<PY
import sys, os, time
import numpy as np
sys.path.insert(0, "$ROOT")
import torch
import cyberether_amd.jetstream as js
for n, b in ((32768, 64), (131072, 32), (20000, 64)):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))).astype(np.complex64)
    src = js.Tensor.from_numpy(x, batch=0, sample=1)
    eng = js.SpectrumEngine(src, enable_scale=True, range_min=-100.0, range_max=0.0)
    rt = js.Runtime(eng.modules, graph=True, fuse=True)
    rt.compute(10, sync=True); torch.cuda.synchronize(); t0 = time.perf_counter(); rt.compute(200, sync=False); rt.synchronize(); torch.cuda.synchronize()
    print("  $v", n, b, round((time.perf_counter() - t0) / 200 * 1e6, 2), "us per cycle")
    rt.destroy()
PY
done; done
cp /tmp/base.so $L
