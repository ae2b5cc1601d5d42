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


@pytest.fixture
def aids(monkeypatch):
    """For the duration of a test: the -DCW_TEST_AIDS build of the library (consent_amd/aids/), in this process and -- through
    LD_LIBRARY_PATH -- in the executables of bin/ it starts.  Only that build reads the test aids (CW_TASK_CAP, CW_DRIVER_DRY, ...), the
    experiment knobs and holds the opt-in kernels; the product library ignores them (csrc/cw_env.h)."""
    from consent_amd import engine

    prev = engine.use_library(engine.AIDS_LIB)
    monkeypatch.setenv("LD_LIBRARY_PATH", os.path.dirname(engine.AIDS_LIB) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    yield engine.AIDS_LIB
    engine._LIB = prev
