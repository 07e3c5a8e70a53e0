import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def js():
    """The product's host layer; building it is the driver's build() step."""
    lib = os.path.join(ROOT, "cyberether_amd", "lib", "libjetstream_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    import cyberether_amd.jetstream as mod
    return mod


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as mod
    mod.build()
    return mod
