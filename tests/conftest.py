import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the native artefacts once if they are missing (CPU-only: hipcc cross-compiles)."""
    from genomicsbench_amd import build
    need = [os.path.join(ROOT, "genomicsbench_amd", "libgbx.so"),
            os.path.join(ROOT, "genomicsbench_amd", "libgbx_datagen.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        build.build_all()
    yield


def has_gpu():
    try:
        from genomicsbench_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False
