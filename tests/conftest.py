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


@pytest.fixture
def devlib(built):
    """Objects created inside the test live on the DEVELOPER build (build/libcdae_hip_dev.so: the same sources with -DCDAE_DEVELOPER),
    the only library that reads the developer environment switches (CDAE_SORT_TILE, CDAE_GEMM1_TILED, ...).  A test that compares a
    switched path with the default one runs BOTH sides on it; the shipped library's indifference to those variables is asserted by
    tests/test_gpu_parity.py::test_the_shipped_library_reads_no_developer_switch."""
    import cdae_amd
    with cdae_amd.developer_library() as lib:
        yield lib
