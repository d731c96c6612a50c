"""Loader for libxrslam_hip.so.  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# XRSLAM_HIP_LIB selects an instrumented build variant of the same library (csrc/build.sh XR_VARIANT)
LIB_PATH = os.environ.get("XRSLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libxrslam_hip.so")

XRHIP_OK = 0
XRHIP_ENODEVICE = -1
XRHIP_EINVAL = -2
XRHIP_EHIP = -3
XRHIP_ENOMEM = -4
XRHIP_EOVERFLOW = -5
XRHIP_ESTATE = -6


class XrhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("xrslam_hip error %d: %s" % (code, msg))
        self.code = code


def build():
    """Compile the gfx950 library in-tree (hipcc cross-compiles without a GPU).  build.sh is incremental: it recompiles
    only the translation units whose sources or headers are newer than their objects."""
    subprocess.check_call([os.path.join(_HERE, "csrc", "build.sh")])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XrhipError(XRHIP_ENODEVICE,
                             "libxrslam_hip.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.xrhip_last_error.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise XrhipError(rc, lib().xrhip_last_error().decode("utf-8", "replace"))
    return rc


def kernel_revision():
    f = lib().xrhip_kernel_revision
    f.restype = C.c_char_p
    return f().decode()


def device_count():
    return lib().xrhip_device_count()


def set_device(i):
    check(lib().xrhip_set_device(int(i)))
