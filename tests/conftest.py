import os
import sys

import pytest

# Tests that do not name a datapath are the tests of the exact-fp32 ANCHOR (their tolerances are the fp32 datapath's); the package
# default a user gets is fp16x3 (tests/test_host_cpu.py::test_package_defaults checks that in a clean subprocess).
os.environ.setdefault("NERF_PRECISION", "fp32")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test (CPU tests call its host-only entry points)."""
    import __graft_entry__
    __graft_entry__.build()
