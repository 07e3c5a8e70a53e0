"""GPU: the differential matrix of tests/reference_matrix.py -- 288 (module, config, input layout) tuples written after the reference's
own module_tests.cc files -- driven THROUGH THE REFERENCE on DeviceType::HIP: Registry::BuildModule(type, HIP, NATIVE) ->
Module::create (the reference's own validation on device tensors built with its own slice / permute / expandDims) ->
Runtime(HIP)::compute (integration/device_hip/, linked into oracle/_ref/libref_jetstream_devhip.so).  For every case the
reference-on-HIP must take the decision the reference-on-CPU took (frozen by tools/make_reference_matrix.py) and, where it accepts,
leave the same bits in its output tensor (read back from HBM) with the same signal axes."""
import numpy as np
import pytest

import reference_matrix as rm
from oracle import ref_jetstream as rj
from util import assert_bit_equal

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not rj.device_hip_library_available(), reason="oracle/_ref/libref_jetstream_devhip.so not built")]

# Decisions the HIP device takes differently, each for a stated reason
KNOWN = {}


@pytest.fixture(scope="module", autouse=True)
def reference_with_the_hip_device(js):
    rj.use_device_hip_library()
    rj.hip_runtime_configure(True, 0)
    yield


@pytest.mark.parametrize("name", rm.names())
def test_reference_on_hip_takes_its_own_cpu_decision_and_output(name):
    rec = rm.load()[name]
    c = rm.by_name(name)
    if name in KNOWN:
        pytest.xfail(KNOWN[name])
    code, outs, axes = rm.run_reference(c, device="hip")
    assert (code == 0) == (rec["code"] == 0), f"{name} ({rec['cite']}): CPU device Result {rec['code']}, HIP device Result {code}"
    if code != 0:
        return
    assert axes == rec["axes"], (name, axes, rec["axes"])
    for k, (got, want) in enumerate(zip(outs, rec["outs"])):
        assert_bit_equal(np.asarray(got), want, f"{name} cycle {k} ({rec['cite']})")
