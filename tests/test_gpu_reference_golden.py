"""GPU: the HIP path (through the C ABI) against the outputs the REFERENCE ITSELF produced -- every frozen case of
tests/golden/reference_vectors.npz, bit for bit, per-module and fused / graph-captured runtimes alike -- plus the
assertions of the reference's own tests on the device's outputs (filter_engine/block_tests.cc:584-724,
filter/block_tests.cc:343-400, fm/module_tests.cc:204-483; transcribed in tests/reference_cases.py)."""
import numpy as np
import pytest

import reference_cases as rc
from util import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [{}, {"graph": True, "fuse": True}], ids=["per_module", "fused_graph"])
@pytest.mark.parametrize("name", rc.names())
def test_hip_reproduces_the_reference_outputs(js, name, flags):
    case = rc.load()[name]
    got = rc.run_hip(js, case, **flags)
    for c, (g, want) in enumerate(zip(got, case["outs"])):
        assert_bit_equal(np.asarray(g), want, f"{name} cycle {c} ({case['source']})")
    rc.expectations(name, got)


def test_filter_engine_flowgraph_node(js):
    """The filter_engine block as a flowgraph node fed by a filter_taps node (the reference's own split of `filter`)."""
    from cyberether_amd.flowgraph import Flowgraph
    case = rc.load()["filter_block_three_heads"]
    p = case["params"]
    doc = f"""
graph:
  - name: iq
    module: soapy
    config: {{numberOfBatches: 3, numberOfTimeSamples: 1950, sampleRate: {p['sampleRate']}}}
  - name: taps
    module: filter_taps
    config: {{sampleRate: {p['sampleRate']}, bandwidth: {p['bandwidth']}, center: {p['center']}, taps: {p['taps']}, heads: 3}}
  - name: eng
    module: filter_engine
    input: {{signal: '${{graph.iq.output.signal}}', filter: '${{graph.taps.output.coeffs}}'}}
"""
    fg = Flowgraph(doc, ring_slots=1)
    fg.feed("iq", case["ins"][0])
    rt = fg.runtime(graph=False, fuse=True)
    rt.compute(1)
    assert_bit_equal(fg.output("eng", "buffer").numpy(), case["outs"][0], "filter_taps -> filter_engine == filter block")
    rt.destroy()
