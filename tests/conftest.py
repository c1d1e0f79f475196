import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """The process-wide device context (GPU tests only)."""
    from ipde_amd.device import get_context
    return get_context()


@pytest.fixture(autouse=True)
def _library_options_as_found(request):
    """The library's tuning knobs are state of the shared device context: whatever a GPU test sets
    (or leaves set by failing half-way) is put back before the next test runs."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from ipde_amd import device
    snap = device.snapshot_options()
    yield
    device.restore_options(snap)
