"""Two more oracle stages pinned to the reference ITSELF rather than to a careful restatement: Backend::ApproxLog10
(include/jetstream/backend/devices/cpu/helpers.hh:59-74 -- what Amplitude calls; the reference's own tests only hold it
to 0.5 dB) and the waterfall ring arithmetic (waterfall/ring_state.hh:16-56), both header-inline and compiled in place
by oracle/Makefile into oracle/_ref/libref_helpers.so.  The restatement must return the same bits on EVERY float and
the same plans on a (writeIndex, batches, height) grid."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.skipif(not oracle.have_ref_helpers(), reason="oracle/_ref/libref_helpers.so not built (no reference tree / fmt headers)")


def test_approx_log10_equals_the_reference_on_every_float():
    ref, lib = oracle.ref_helpers(), oracle.lib()
    f32p = C.POINTER(C.c_float)
    for fn in (ref.ref_approx_log10_bits, lib.jst_oracle_approx_log10_bits):
        fn.restype, fn.argtypes = None, [C.c_uint32, C.c_uint64, f32p]
    chunk = 1 << 24
    a, b = np.empty(chunk, np.float32), np.empty(chunk, np.float32)
    checked = 0
    # every non-negative bit pattern (zero, subnormals, normals, inf, NaNs), then a band of negative ones (the function
    # takes fabs first)
    starts = list(range(0, 1 << 31, chunk)) + [0x80000000, 0xbf000000, 0xff000000]
    for first in starts:
        ref.ref_approx_log10_bits(first, chunk, a.ctypes.data_as(f32p))
        lib.jst_oracle_approx_log10_bits(first, chunk, b.ctypes.data_as(f32p))
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"first mismatch at bits {first + int(np.flatnonzero(~same)[0]):#x}"
        checked += chunk
    assert checked == (1 << 31) + 3 * chunk


def test_waterfall_ring_arithmetic_equals_the_reference_on_a_grid():
    ref = oracle.ref_helpers()
    u64p = C.POINTER(C.c_uint64)
    ref.ref_waterfall_plan.restype, ref.ref_waterfall_plan.argtypes = None, [C.c_uint64, C.c_uint64, C.c_uint64, u64p]
    ref.ref_waterfall_advance.restype, ref.ref_waterfall_advance.argtypes = None, [u64p, C.c_uint64, C.c_uint64, u64p]
    rng = np.random.default_rng(7)
    heights = [1, 2, 3, 7, 16, 255, 256, 512, 2048]
    for h in heights:
        for b in sorted({1, 2, 3, h - 1 if h > 1 else 1, h, h + 1, 2 * h, 2 * h + 3, 5 * h + 1, 1024}):
            for w in sorted({0, 1, h // 2, h - 1} | set(int(v) for v in rng.integers(0, h, 4))):
                out = (C.c_uint64 * 3)()
                ref.ref_waterfall_plan(w, b, h, out)
                assert tuple(oracle.waterfall_plan(w, b, h)) == tuple(out), (w, b, h)
    for h in heights:
        state_ref = (C.c_uint64 * 2)(0, 0)
        state = (0, 0)
        for step in range(40):
            b = int(rng.integers(1, 3 * h + 2))
            dirty = (C.c_uint64 * 3)()
            ref.ref_waterfall_advance(state_ref, b, h, dirty)
            state = oracle.waterfall_advance(state, b, h)
            assert tuple(state) == (state_ref[0], state_ref[1]), (h, step)
            assert tuple(oracle.waterfall_dirty_plan(state, h)) == tuple(dirty), (h, step)
            if step % 7 == 6:   # the module clears the dirty rows after presenting (ring_state.hh:52-54)
                state_ref[1] = 0
                state = (state[0], 0)
