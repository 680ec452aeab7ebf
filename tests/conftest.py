import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle) and, when the extension is absent and hipcc is present, the extension.
    On the GPU box both arrive prebuilt with the snapshot."""
    from oracle import oracle_ffi
    oracle_ffi.build()
    import bbtools_amd
    if not os.path.exists(bbtools_amd.lib_path()):
        bbtools_amd.build()
    yield
