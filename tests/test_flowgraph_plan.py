"""Flowgraph loader, host side (no device): parsing, dependency ordering, block classification.
When the reference checkout is present (this container only) its own example flowgraphs are
parsed too -- the files are read in place, never copied."""
import glob
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = os.path.join(HERE, "flowgraphs")


def _plan(path):
    from cyberether_amd.flowgraph import Flowgraph
    return Flowgraph(path, instantiate=False)


def test_fixture_console_order_and_classification():
    fg = _plan(os.path.join(FIXTURES, "spectrum_console.yml"))
    names = [p["name"] for p in fg.plan]
    assert fg.dropped == ["readme"]
    assert set(names) == {"rng", "win", "sdr", "inv", "mul", "fft", "amp", "plot", "wtf"}
    pos = {n: i for i, n in enumerate(names)}
    for a, b in (("win", "inv"), ("inv", "mul"), ("sdr", "mul"), ("mul", "fft"), ("fft", "amp"),
                 ("amp", "rng"), ("rng", "plot"), ("rng", "wtf")):
        assert pos[a] < pos[b], (a, b)
    assert all(p["status"] == "ok" for p in fg.plan)
    assert fg.plan[pos["mul"]]["inputs"] == {"a": "sdr.signal", "b": "inv.signal"}


def test_fixture_fm_skips_host_sink():
    fg = _plan(os.path.join(FIXTURES, "two_station_fm.yml"))
    status = {p["name"]: p["status"] for p in fg.plan}
    assert status["audio"] == "skipped" and status["flt"] == "ok" and status["station"] == "ok"
    assert sum(v == "ok" for v in status.values()) == 8


def test_bad_graphs_are_rejected():
    from cyberether_amd.flowgraph import Flowgraph, FlowgraphError
    with pytest.raises(FlowgraphError, match="no 'graph:' section"):
        Flowgraph("title: x\n", instantiate=False)
    cyc = ("graph:\n  - {name: a, module: invert, input: {signal: '${graph.b.output.signal}'}}\n"
           "  - {name: b, module: invert, input: {signal: '${graph.a.output.signal}'}}\n")
    with pytest.raises(FlowgraphError, match="unresolved or cyclic"):
        Flowgraph(cyc, instantiate=False)
    with pytest.raises(FlowgraphError, match="cannot parse input reference"):
        Flowgraph("graph:\n  - {name: a, module: invert, input: {signal: 'b.signal'}}\n", instantiate=False)


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples/flowgraphs"),
                    reason="reference checkout not present (GPU box)")
def test_reference_examples_parse_in_place():
    seen = {}
    for path in sorted(glob.glob("/root/reference/examples/flowgraphs/*.yml")):
        fg = _plan(path)
        seen[os.path.basename(path)] = {p["block"]: p["status"] for p in fg.plan}
    assert len(seen) >= 7
    unsupported = {b for blocks in seen.values() for b, s in blocks.items() if s == "unsupported"}
    assert unsupported == set(), unsupported  # every block is built, or knowingly skipped (audio, adsb)
    skipped = {b for blocks in seen.values() for b, s in blocks.items() if s == "skipped"}
    assert skipped <= {"audio", "adsb"}
