import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lib_built():
    """Make sure libcfm_gfx950.so exists (hipcc cross-compiles without a GPU)."""
    import cfm_amd  # noqa: F401
    from cfm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib
