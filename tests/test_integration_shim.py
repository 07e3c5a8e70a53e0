"""CPU: the reference-side bindings INTEGRATION.md shows (sections 2 and 5) are real code: every ```cpp block of the
document is extracted and compiled (`g++ -std=c++20 -fsyntax-only`) against the reference's OWN headers
(/root/reference/include, the module's source directory for its module_impl.hh, oracle/ref_shim for the config / fmt
stand-ins the reference's meson build would generate) and against this repo's include/jetstream_hip.h.  Skipped where
the reference tree is absent (the GPU box)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "jetstream", "registry.hh")),
                                reason="reference tree not present")


def blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = []
    for body in re.findall(r"```cpp\n(.*?)```", text, re.S):
        first = body.splitlines()[0]
        m = re.search(r"\b(src/[\w/.]+\.cc)\b", first)
        assert m, f"a cpp block of INTEGRATION.md must start with the path it would have in the reference tree: {first!r}"
        out.append((m.group(1), body))
    return out


def torch_include():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "include")


@pytest.mark.parametrize("path,body", blocks(), ids=[p for p, _ in blocks()])
def test_binding_compiles_against_the_reference_headers(tmp_path, path, body):
    src = tmp_path / os.path.basename(path)
    src.write_text(body)
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-DFMT_HEADER_ONLY=1",
           "-I" + os.path.join(ROOT, "oracle", "ref_shim"), "-I" + os.path.join(REF, "include"),
           "-I" + os.path.join(REF, os.path.dirname(path)), "-I" + os.path.join(REF, "src"),
           "-I" + torch_include(), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "integration", "mi355x_provider"), str(src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_the_document_shows_the_fft_unit_as_it_is_and_the_producer():
    """Section 2's block IS integration/mi355x_provider/fft.cc (the units that are linked into the compiled reference and run on
    the GPU: tests/test_gpu_reference_drives_library.py; all nine compile: tests/test_integration_device_hip.py)."""
    found = dict(blocks())
    assert any("io/soapy/" in p for p in found)
    key = next(p for p in found if "dsp/fft/" in p)
    assert found[key] == open(os.path.join(ROOT, "integration", "mi355x_provider", "fft.cc")).read()
