import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from texturefusion_amd import capi
        return capi.lib().tf_device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu_required():
    """-m gpu tests must run the HIP path; on a box without a GPU they fail loudly, never skip to a fallback."""
    n = _gpu_count()
    if n <= 0:
        pytest.fail("no HIP device visible: gpu-marked tests need the MI355X (there is no CPU fallback)")
    return n
