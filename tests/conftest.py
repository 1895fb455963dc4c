import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# every accumulate of the GPU tests is followed by uvcgpu_region_check_presence (uvc_amd/region.py): a plane cell outside the symbols the
# scoring gather sums would change results silently
os.environ.setdefault("UVCGPU_CHECK_PRESENCE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure only)."""
    from uvc_amd import _ffi
    path = __import__("oracle").library_path()
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so")])
    return _ffi.Lib(path, "uvc_oracle_")


@pytest.fixture(scope="session")
def gpu_lib():
    from uvc_amd import region
    lib = region.gpu_lib()
    rc = lib.dll.uvcgpu_init(0)
    assert rc == 0, lib.last_error()
    return lib
