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


@pytest.fixture(autouse=True)
def _summaries_cover_the_voxels(request, monkeypatch):
    """Every GPU test doubles as a check of the filter summaries the voxel writers maintain (tf_check_summaries): when
    a test closes a volume, no chunk's summary may lack a class its voxels hold."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from texturefusion_amd import capi
    real_close = capi.Volume.close
    seen = []
    nbrs = []

    def close(self):
        if getattr(self, "h", None):
            try:
                seen.append(self.check_summaries())
                nbrs.append(self.check_neighbours())
            except capi.TFError:
                pass  # a handle a test has deliberately left in an error state
        real_close(self)

    monkeypatch.setattr(capi.Volume, "close", close)
    yield
    for n, missing, stale in seen:
        assert missing == 0, "filter summaries lack classes in %d of %d chunks" % (missing, n)
    for c in nbrs:  # ... and of the neighbour table the filter / patch stage read instead of probing the hash
        assert c[2] == 0 and c[4] == 0 and c[5] == 0, "neighbour table disagrees with the chunk hash: %s" % (c.tolist(),)
