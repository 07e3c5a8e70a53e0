"""GPU: the product against the reference over the differential matrix of tests/reference_matrix.py -- 288 (module, config,
input layout) tuples written after the reference's own module tests (dense / batched / multi-head / rank-3 / strided / offset
layouts, every sample type of the cast, every waveform of the signal generator, broadcast forms of multiply, state across
submissions, and the malformed inputs of the validation sections).  For every case the HIP path, through ctypes -> C ABI, must
take the reference's DECISION (accept / reject: the reference's Result code frozen by tools/make_reference_matrix.py from the
reference compiled in place) and, where it accepts, produce the reference's output BIT FOR BIT with the same signal axes."""
import numpy as np
import pytest

import reference_matrix as rm
from util import assert_bit_equal

pytestmark = pytest.mark.gpu

# Known, deliberate differences (each one a decision documented where it is taken, none on the BASELINE path)
KNOWN = {}


@pytest.mark.parametrize("name", rm.names())
def test_hip_takes_the_references_decision_and_matches_its_output(js, name):
    rec = rm.load()[name]
    c = rm.by_name(name)
    if name in KNOWN:
        pytest.xfail(KNOWN[name])
    ok, outs, axes = rm.run_hip(js, c)
    ref_says, hip_says = ("accepts" if rec["code"] == 0 else "rejects"), ("accepts" if ok else "rejects")
    assert ok == (rec["code"] == 0), f"{name} ({rec['cite']}): the reference {ref_says} this input, the HIP path {hip_says}"
    if not ok:
        return
    assert axes == rec["axes"], (name, axes, rec["axes"])
    for k, (got, want) in enumerate(zip(outs, rec["outs"])):
        assert_bit_equal(np.asarray(got), want, f"{name} cycle {k} ({rec['cite']})")
