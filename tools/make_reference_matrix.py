#!/usr/bin/env python3
"""Runs every case of tests/reference_matrix.py on the REFERENCE compiled in place (oracle/_ref/libref_jetstream.so) and freezes
(Result code, outputs per compute cycle, output signal axes) into tests/golden/reference_matrix.npz, so that the GPU box -- where
/root/reference does not exist -- can hold the HIP path against the reference's own decisions and outputs case by case
(tests/test_gpu_reference_matrix.py).  Run in the build container:  python tools/make_reference_matrix.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import reference_matrix as rm  # noqa: E402

manifest, arrays = {}, {}
accepted = rejected = 0
for c in rm.CASES:
    code, outs, axes = rm.run_reference(c)
    manifest[c["name"]] = {"code": int(code), "cycles": c["cycles"], "axes": axes, "cite": c["cite"], "module": c["module"]}
    if code == 0:
        accepted += 1
        for k, o in enumerate(outs):
            arrays[f"{c['name']}/out{k}"] = o
    else:
        rejected += 1
np.savez_compressed(rm._PATH, manifest=np.frombuffer(json.dumps(manifest, sort_keys=True).encode(), np.uint8), **arrays)
print(f"wrote {rm._PATH}: {len(manifest)} cases ({accepted} accepted, {rejected} rejected by the reference), {os.path.getsize(rm._PATH) / 1024:.0f} KiB")
