import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "orb_slam_amd", "liborbx.so"), os.path.join(ROOT, "orb_slam_amd", "libsynthframes.so"),
            os.path.join(ROOT, "oracle", "liborb_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "all"])


_ensure_built()


@pytest.fixture(scope="session")
def gpu_extractor_factory():
    """Creates product extractors; fails loudly (no skip, no fallback) when the HIP path is unusable."""
    from orb_slam_amd import capi
    made = []

    def make(**kw):
        ex = capi.ORBextractor(**kw)
        made.append(ex)
        return ex

    yield make
    for ex in made:
        ex.close()
