"""The bit-exact float parity is stated against one libm (glibc 2.35 x86-64, FMA IFUNC variants): the pin file exists,
the probe runs, and the tolerance fallback of assert_bit_equal does what its docstring says on a host that differs."""
import json
import os
import warnings

import numpy as np
import pytest

import util


def test_pin_file_and_probe():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libm_pin.json")
    pin = json.load(open(path))
    assert pin["libc"].startswith("glibc 2.35") and pin["machine"] == "x86_64"
    assert set(pin["functions"]) == {"tanhf", "expm1f", "sinf", "cosf", "atan2f"}
    assert isinstance(util.libm_pinned(), bool)


def test_fallback_only_when_libm_differs(monkeypatch):
    ref = np.linspace(0.0, 1.0, 1000, dtype=np.float32)
    ulp = ref.copy()
    ulp.view(np.uint32)[500] += 1
    monkeypatch.setattr(util, "_LIBM_PINNED", True)
    with pytest.raises(AssertionError):
        util.assert_bit_equal(ulp, ref, "pinned host: one ulp is a failure")
    monkeypatch.setattr(util, "_LIBM_PINNED", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        util.assert_bit_equal(ulp, ref, "other libm: judged at 1e-5 of the peak")
    assert any("libm differs" in str(x.message) for x in w)
    gross = ref.copy()
    gross[10] += 1e-3
    with pytest.raises(AssertionError):
        util.assert_bit_equal(gross, ref, "outside the tolerance fails on any host")
    ints = np.arange(10, dtype=np.uint32)
    with pytest.raises(AssertionError):
        util.assert_bit_equal(ints + 1, ints, "integer work is always bit-exact")
