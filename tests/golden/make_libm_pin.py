"""Records what the host libm returns for a fixed argument set (tanhf, sinf, cosf, atan2f, expm1f): the bit-exact
float parity of Range and FM is parity with THIS libm (glibc 2.35, x86-64, the FMA IFUNC variants of sinf/cosf) --
the reference calls the C library for them (range/module_impl_native_cpu.cc:67-82, fm/module_impl_native_cpu.cc:43-174).
tests/util.py compares the running host against this file; on a host whose libm answers differently the bit-exact
assertions fall back to BASELINE.json's tolerance (1e-5 of the peak) and say "libm differs".
    python tests/golden/make_libm_pin.py      (run in the build container; writes tests/golden/libm_pin.json)"""
import ctypes as C
import json
import os
import platform

import numpy as np


def probe():
    libm = C.CDLL("libm.so.6")
    rng = np.random.default_rng(20260924)
    out = {}
    for name, lo, hi in (("tanhf", -8.0, 8.0), ("expm1f", -17.0, 17.0), ("sinf", -200.0, 200.0), ("cosf", -200.0, 200.0)):
        fn = getattr(libm, name)
        fn.restype, fn.argtypes = C.c_float, [C.c_float]
        args = rng.uniform(lo, hi, 96).astype(np.float32)
        res = np.array([fn(float(a)) for a in args], np.float32)
        out[name] = {"args": args.view(np.uint32).tolist(), "bits": res.view(np.uint32).tolist()}
    fn = libm.atan2f
    fn.restype, fn.argtypes = C.c_float, [C.c_float, C.c_float]
    ya = rng.uniform(-4.0, 4.0, 96).astype(np.float32)
    xa = rng.uniform(-4.0, 4.0, 96).astype(np.float32)
    res = np.array([fn(float(y), float(x)) for y, x in zip(ya, xa)], np.float32)
    out["atan2f"] = {"args": ya.view(np.uint32).tolist(), "args2": xa.view(np.uint32).tolist(),
                     "bits": res.view(np.uint32).tolist()}
    return out


if __name__ == "__main__":
    data = {"libc": " ".join(platform.libc_ver()), "machine": platform.machine(), "functions": probe()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libm_pin.json")
    with open(path, "w") as f:
        json.dump(data, f)
    print(path, data["libc"], data["machine"])
