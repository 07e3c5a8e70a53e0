"""CPU: the differential matrix over the reference's module tests (tests/reference_matrix.py) is frozen completely -- every
case has the reference's decision (Result code) and, where it accepted the input, its outputs -- and, where the compiled
reference is present, re-running a sample of the cases reproduces the frozen bits (the freeze is what the reference does, not a
stale file).  The GPU half is tests/test_gpu_reference_matrix.py."""
import numpy as np
import pytest

import reference_matrix as rm


def test_every_case_is_frozen_with_a_citation():
    frozen = rm.load()
    assert set(frozen) == set(rm.names())
    assert len(frozen) >= 250
    assert sum(1 for r in frozen.values() if r["code"] != 0) >= 40      # the validation sections are in
    for name, rec in frozen.items():
        assert "module_tests.cc" in rec["cite"], name
        if rec["code"] == 0:
            assert len(rec["outs"]) == rm.by_name(name)["cycles"]
    modules = {rm.by_name(n)["module"] for n in frozen}
    assert {"fft", "cast", "signal_generator", "amplitude", "multiply", "range", "invert", "window", "pad", "unpad", "fold",
            "arithmetic", "phase_correction", "overlap_add", "agc", "am", "add", "multiply_constant"} <= modules


def _reference_available():
    from oracle import ref_jetstream as rj
    return rj.available()


@pytest.mark.skipif(not _reference_available(), reason="oracle/_ref/libref_jetstream.so not built")
@pytest.mark.parametrize("name", rm.names()[::7])
def test_the_reference_reproduces_its_frozen_results(name):
    rec = rm.load()[name]
    code, outs, axes = rm.run_reference(rm.by_name(name))
    assert code == rec["code"]
    if code == 0:
        assert axes == rec["axes"]
        for k, (got, want) in enumerate(zip(outs, rec["outs"])):
            assert got.shape == want.shape and got.dtype == want.dtype
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint8), np.ascontiguousarray(want).view(np.uint8)), (name, k)
