"""pytest configuration: `-m gpu` tests need an MI355X and call through the C ABI; everything else runs on CPU."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def rt():
    """One sc_runtime for the whole GPU session.  No fallback: without a GPU this raises."""
    from semcode_amd import _native

    runtime = _native.Runtime(device=0)
    yield runtime
    runtime.close()
