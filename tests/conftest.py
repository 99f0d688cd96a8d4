import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: it must be loaded before libpolypolish_hip.so pulls in the
    import torch  # noqa: F401  system one, otherwise torch later finds "no ROCm-capable device"
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The C oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc
