import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def host_mirror():
    """tests/csrc/host_mirror.cc compiled with g++ (CPU build of the device data structures)."""
    import ctypes
    src = os.path.join(ROOT, "tests", "csrc", "host_mirror.cc")
    so = os.path.join(ROOT, "tests", "csrc", "libhost_mirror.so")
    hdrs = [os.path.join(ROOT, "rlap_amd", "csrc", h) for h in ("rlap_core.h", "rlap_flow.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-msse4.2", "-mavx", "-fPIC", "-shared", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.mirror_approx_chol.restype = ctypes.c_int
    return lib
