"""x264vfw_amd — MI355X-native H.264 encode hot path behind the x264vfw / libx264 C API.

The product is the C-ABI shared library ``libx264gpu.so`` (hand-written HIP for gfx950, see
``include/x264gpu.h``).  This package is only the Python-side loader + thin mirrors used by the tests
and by ``bench.py``.  There is no CPU fallback: importing :mod:`x264vfw_amd.lib` raises if the HIP
library has not been built.
"""
__all__ = ["lib"]
