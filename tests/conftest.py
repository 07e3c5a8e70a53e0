import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """Which float-comparison policy this box takes (tests/util.py: assert_bit_equal): with the pinned libm (glibc 2.35
    answers of tests/golden/libm_pin.json) every float is compared bit for bit; on any other libm a float mismatch is
    re-judged at 1e-5 of the peak with a warning.  Recorded in the run's header so the driver's log says which."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import libm_pinned
        pinned = libm_pinned()
    except Exception as exc:  # the header must never break a run
        return f"jetstream-hip: libm pin check failed ({exc!r})"
    ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_jetstream.so"))
    return (f"jetstream-hip: libm_pinned={pinned} (float parity {'bit for bit' if pinned else 'at 1e-5 of the peak, with warnings'}); "
            f"compiled reference (oracle/_ref/libref_jetstream.so) {'present' if ref else 'absent: frozen vectors only'}")


@pytest.fixture(scope="session")
def js():
    """The product's host layer; building it is the driver's build() step."""
    lib = os.path.join(ROOT, "cyberether_amd", "lib", "libjetstream_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    import cyberether_amd.jetstream as mod
    return mod


@pytest.fixture
def switch(js):
    """switch(name, value): flips one of the library's A/B switches (jst_debug_set) for this test; all are unset afterwards."""
    touched = []

    def set_(name, value):
        touched.append(name)
        js.debug_set(name, value)
    yield set_
    for name in touched:
        js.debug_set(name, None)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as mod
    mod.build()
    return mod
