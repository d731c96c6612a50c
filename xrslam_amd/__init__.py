"""xrslam_amd -- MI355X-native (gfx950) implementation of the XRSLAM per-frame hot path.

The product is the C-ABI shared library ``xrslam_amd/lib/libxrslam_hip.so``
(sources in ``xrslam_amd/csrc``, header ``include/xrslam_hip.h``).  The Python
modules here are thin ctypes mirrors of the reference's plug-point interfaces
used by the tests and by ``bench.py``; they contain no arithmetic and never
fall back to a CPU implementation.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
