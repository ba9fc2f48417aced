import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native artefacts: the gfx950 C-ABI library (cross-compiled, no GPU needed) and the CPU oracle."""
    import __graft_entry__ as g
    g.build()
    return g
