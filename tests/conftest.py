import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The HIP library, initialised on device 0.  Fails (does not skip) when the
    extension is missing on a GPU box: the product has no CPU fallback."""
    from plinopt_amd import capi
    L = capi.lib()
    capi.check(L.plo_init(0))
    return L
