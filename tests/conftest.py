import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle/) and, when stale and hipcc is present, the product library."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from consent_amd import _build

    if os.path.exists("/opt/rocm/bin/hipcc"):
        _build.build(verbose=False)  # rebuilds the library and the bin/ executables only when stale
    yield


def gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
