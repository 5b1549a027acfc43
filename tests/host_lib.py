"""tests' entry to the B1 binding (x264vfw_amd/host_api.py).  With X264_HOST_STUB set the product's host sources are linked against the
stand-in device library (tests/stub/: the oracle behind the B3 ABI) — only in a process that never loads the real libx264gpu.so."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("X264_HOST_STUB"):
    _b = os.path.join(ROOT, "tests", "stub", "_build")
    if not os.path.exists(os.path.join(_b, "libx264gpu_host.so")):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "stub")])
    os.environ["X264GPU_HOST_LIB"] = os.path.join(_b, "libx264gpu_host.so")
from x264vfw_amd.host_api import *  # noqa: E402,F401,F403
from x264vfw_amd.host_api import H  # noqa: E402,F401
