"""CPU: the branch-free tanhf the GPU Range epilogue uses (kernels/libm_float.hh) compiled for
the host and swept against this host's libm tanhf -- every bit must agree.  (The device build of
the same text is swept on the GPU by tests/test_gpu_elementwise.py.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    out = tmp_path_factory.mktemp("libm") / "libm_check.so"
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-fPIC", "-shared",
                           os.path.join(HERE, "native", "libm_check.cc"), "-o", str(out), "-lm"])
    lib = C.CDLL(str(out))
    lib.jst_tanhf_mismatches.restype = C.c_uint64
    lib.jst_tanhf_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.c_int]
    lib.jst_tanhf_branchy.restype = C.c_float
    lib.jst_tanhf_branchy.argtypes = [C.c_float]
    lib.jst_tanhf_select.restype = C.c_float
    lib.jst_tanhf_select.argtypes = [C.c_float]
    lib.jst_trig_mismatches.restype = C.c_uint64
    lib.jst_trig_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.c_int]
    lib.jst_atan2f_mismatches.restype = C.c_uint64
    lib.jst_atan2f_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    return lib


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_every_257th_float(checker, variant):
    first = C.c_uint32(0)
    n = (1 << 32) // 257
    bad = checker.jst_tanhf_mismatches(0, 257, n, C.byref(first), variant)
    assert bad == 0, f"{bad} mismatches, first at bits {first.value:#x}"


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_dense_where_range_lives(checker, variant):
    # the Range module feeds 4*(normalized-0.5): |x| mostly in [2^-10, 8) -> every float there
    first = C.c_uint32(0)
    lo, hi = np.float32(2.0 ** -10).view(np.uint32), np.float32(8.0).view(np.uint32)
    for sign in (0, 0x80000000):
        bad = checker.jst_tanhf_mismatches(int(lo) | sign, 3, (int(hi) - int(lo)) // 3, C.byref(first), variant)
        assert bad == 0, f"{bad} mismatches, first at bits {first.value:#x}"


@pytest.mark.parametrize("variant", [0, 1])
def test_branch_boundaries(checker, variant):
    edges = [0x24000000, 0x33000000, 0x3eb17218, 0x3F851592, 0x3f800000, 0x41b00000, 0x7f800000,
             0x4195b844, 0x3e800000, 0x3f000000]
    first = C.c_uint32(0)
    for e in edges:
        for sign in (0, 0x80000000):
            start = ((e >> 1) - 64) | (sign >> 1)  # expm1 sees 2|x|: probe around e/2 ...
            assert checker.jst_tanhf_mismatches((e - 64) | sign, 1, 128, C.byref(first), variant) == 0, hex(first.value)
            assert checker.jst_tanhf_mismatches((e - 0x00800000 - 64) | sign, 1, 128, C.byref(first), variant) == 0
    fn = checker.jst_tanhf_branchy if variant else checker.jst_tanhf_select
    assert fn(0.0) == 0.0 and np.signbit(np.float32(fn(-0.0)))
    assert fn(float("inf")) == 1.0 and fn(float("-inf")) == -1.0
    assert np.isnan(fn(float("nan")))


def test_main_path_form_on_every_float_of_its_domain(checker):
    """libm_tanhf_main (the straight-line form the fused epilogue runs) against libm.so.6 on EVERY float it
    answers itself -- 2^-26 <= |x| < 7.5, both signs, plus a margin into the bail-out region: the unified
    argument reduction and the merged reconstructions need no patch anywhere."""
    from concurrent.futures import ThreadPoolExecutor
    lo, hi = 0x32800000 - 4096, 0x40f00000 + 4096
    chunk = (hi - lo) // 32 + 1

    def run(job):
        start, count = job
        first = C.c_uint32(0)
        return checker.jst_tanhf_mismatches(start, 1, count, C.byref(first), 2), first.value

    jobs = [((lo + i * chunk) | s, min(chunk, hi - (lo + i * chunk))) for s in (0, 0x80000000) for i in range(32)
            if lo + i * chunk < hi]
    with ThreadPoolExecutor(8) as ex:  # ctypes releases the GIL
        res = list(ex.map(run, jobs))
    bad = sum(r[0] for r in res)
    assert bad == 0, f"{bad} mismatches, first at bits {[hex(r[1]) for r in res if r[0]][:4]}"


@pytest.mark.parametrize("which,name", [(0, "sinf"), (1, "cosf"), (2, "atanf")])
def test_fm_trig_restatements_on_every_float(checker, which, name):
    """libm_sinf / libm_cosf / libm_atanf (kernels/libm_float.hh, what the FM kernels call) against this host's
    libm.so.6 on ALL 2^32 floats.  sinf / cosf follow the `_fma` build libm selects on CPUs with FMA + AVX2 (every
    current x86 server): on a CPU without FMA libm runs its non-contracted build, whose double-precision
    intermediate can differ in the last place and, once in ~2^29 arguments, round to the other float -- hence the
    skip there."""
    from concurrent.futures import ThreadPoolExecutor
    if which < 2:
        flags = open("/proc/cpuinfo").read()
        if " fma" not in flags or " avx2" not in flags:
            pytest.skip("host libm runs its non-FMA sinf/cosf build")
    chunk = 1 << 26

    def run(i):
        first = C.c_uint32(0)
        return checker.jst_trig_mismatches(i * chunk, 1, chunk, C.byref(first), which), first.value

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(run, range(64)))
    bad = sum(r[0] for r in res)
    assert bad == 0, f"{name}: {bad} mismatches, first at bits {[hex(r[1]) for r in res if r[0]][:4]}"


def test_fm_atan2f_restatement(checker):
    """libm_atan2f: random bit patterns (every special-case branch), Gaussian IQ at three scales (the discriminator's
    operands) and the axes / infinities / signed zeros explicitly."""
    rng = np.random.default_rng(1)
    n = 1 << 22
    first = C.c_uint64(0)
    cases = [(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32),
              rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32))]
    for scale in (1e-3, 1.0, 1e3):
        cases.append(((rng.standard_normal(n) * scale).astype(np.float32), rng.standard_normal(n).astype(np.float32)))
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38, 3e38, -3e38, 1e-45, 2.0**61, 2.0**-61],
                       np.float32)
    cases.append((np.repeat(special, special.size), np.tile(special, special.size)))
    for ys, xs in cases:
        ys, xs = np.ascontiguousarray(ys), np.ascontiguousarray(xs)
        bad = checker.jst_atan2f_mismatches(ys.ctypes.data, xs.ctypes.data, ys.size, C.byref(first))
        assert bad == 0, f"{bad} mismatches, first at index {first.value}: atan2f({ys[first.value]}, {xs[first.value]})"
