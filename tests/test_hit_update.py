"""CPU: the Spectrogram's hit update in its binade form (kernels/hit_update.hh: one integer multiply-add on the float's
bit pattern per binade instead of up to 64 dependent additions) compiled for the host and compared, bit for bit, with
the additions written out -- every count 0..64 on a dense sweep of starting values in [0, 1] -- and with the reference's
own clamped loop (spectrogram/module_impl_native_cpu.cc:70-77)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    out = tmp_path_factory.mktemp("hits") / "hits_check.so"
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-fPIC", "-shared",
                           os.path.join(HERE, "native", "hits_check.cc"), "-o", str(out)])
    lib = C.CDLL(str(out))
    lib.jst_hits_mismatches.restype = C.c_uint64
    lib.jst_hits_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.jst_hits_reference.restype = C.c_float
    lib.jst_hits_reference.argtypes = [C.c_float, C.c_uint32]
    lib.jst_hits_binade.restype = C.c_float
    lib.jst_hits_binade.argtypes = [C.c_float, C.c_uint32]
    return lib


@pytest.mark.parametrize("first,last,step", [
    (0x00000000, 0x3f800000, 1021),        # every 1021st pattern of [0, 1]: ~1.04 M starts x 65 counts
    (0x3c000000, 0x3f800000, 97),          # denser where the update lives: [2^-7, 1]
    (0x3d7ffff0, 0x3d800010, 1),           # around the tie binade's lower edge 2^-4 ...
    (0x3dffff00, 0x3e000100, 1),           # ... and its upper edge 2^-3
    (0x3effff00, 0x3f000100, 1),           # around 0.5
    (0x3f7ff000, 0x3f800000, 1),           # just below 1
])
def test_binade_form_equals_the_additions(checker, first, last, step):
    bad_w, bad_k = C.c_uint32(0), C.c_uint32(0)
    bad = checker.jst_hits_mismatches(first, last, step, C.byref(bad_w), C.byref(bad_k))
    assert bad == 0, f"{bad} mismatches, first at w bits {bad_w.value:#x}, k = {bad_k.value}"


def test_equals_the_reference_loop_on_random_states(checker):
    rng = np.random.default_rng(5)
    ws = np.concatenate([rng.random(20000, dtype=np.float32), np.float32([0.0, 1.0, 0.36, 0.999, 0.0199999995, 0.98])])
    for w in ws:
        for k in (0, 1, 2, 3, 7, 13, 33, 50, 64):
            a, b = checker.jst_hits_reference(float(w), k), checker.jst_hits_binade(float(w), k)
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (w, k, a, b)
