import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: (re)build it on demand, it is plain C and takes seconds
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".c", ".h", ".cpp"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def gpu():
    """Device handle for -m gpu tests: the HIP library through its C ABI + torch for device buffers."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu test selected but no GPU is visible (no CPU fallback exists)")
    from x264vfw_amd import lib
    assert lib.x264gpu_device_count() >= 1
    return lib
