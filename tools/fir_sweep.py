import os, sys, json, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np
import torch
torch.cuda.set_device(0)
import cyberether_amd.jetstream as js
from bench_configs import timed
rng = np.random.default_rng(0)
b, s, taps, sr, bw = int(os.environ.get("FIR_ROWS", "100")), 159750, 251, 20e6, 2e6
x = (rng.standard_normal((b, s)) + 1j * rng.standard_normal((b, s))).astype(np.complex64)
src = js.Tensor.from_numpy(x, batch=0, sample=1)
ref = None
for tune in sys.argv[1:]:
    os.environ["JST_FIR_TUNE"] = tune
    blk = js.Filter(src, sr, bw, [0.0], taps, 1, provider="fast")
    rt = js.Runtime(blk.modules, graph=True, fuse=True)
    dt = timed(rt, 50, 5)
    y = blk.buffer.numpy()
    if ref is None: ref = y
    print(tune, "%.1f us/cycle" % (dt * 1e6), "max diff vs first", float(np.abs(y - ref).max()), flush=True)
    rt.destroy()
