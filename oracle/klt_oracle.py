"""ctypes front-end for oracle/klt_oracle.c (CPU restatement of the KLT front-end).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle_klt.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle_klt.so"])


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    u8p = C.POINTER(C.c_uint8)
    lib.orc_pyr_create.restype = C.c_void_p
    lib.orc_pyr_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_pyr_destroy.argtypes = [C.c_void_p]
    lib.orc_pyr_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
    lib.orc_pyr_level_dims.restype = C.c_int
    lib.orc_pyr_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_pyr_get_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_pyr_get_level_padded.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_clahe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.orc_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                           C.c_int, C.c_double, C.c_void_p]
    lib.orc_track_keypoints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p]
    lib.orc_harris_response.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    lib.orc_gftt.restype = C.c_int
    lib.orc_gftt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                             C.c_void_p, C.c_void_p]
    lib.orc_detect_keypoints.restype = C.c_int
    lib.orc_detect_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_double, C.c_void_p]
    lib.orc_poisson_select.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_get_threads.restype = C.c_int
    del u8p
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class LkStats(C.Structure):
    _fields_ = [("iters", C.c_longlong), ("templates", C.c_longlong)]


class OracleImage:
    """Mirror of xrslam::extra::OpenCvImage for the hot-path virtuals
    (xrslam/include/xrslam/xrslam.h:137-161)."""

    PAD = 21
    MAX_LEVEL = 3

    def __init__(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        self.h, self.w = gray.shape
        self.raw = gray
        self.image = gray.copy()
        self._pyr = lib().orc_pyr_create(self.w, self.h, self.MAX_LEVEL, self.PAD)

    def __del__(self):
        try:
            if self._pyr:
                lib().orc_pyr_destroy(self._pyr)
                self._pyr = None
        except Exception:
            pass

    def preprocess(self, clip=6.0, tiles_x=8, tiles_y=8):
        out = np.empty_like(self.image)
        lib().orc_preprocess(self._pyr, _p(self.image), self.w, float(clip), tiles_x, tiles_y, _p(out))
        self.image = out

    def level(self, l, padded=False):
        w = C.c_int()
        h = C.c_int()
        lib().orc_pyr_level_dims(self._pyr, l, C.byref(w), C.byref(h))
        if padded:
            ph, pw = h.value + 2 * self.PAD, w.value + 2 * self.PAD
            img = np.empty((ph, pw), np.uint8)
            der = np.empty((ph, pw, 2), np.int16)
            lib().orc_pyr_get_level_padded(self._pyr, l, _p(img), _p(der))
        else:
            img = np.empty((h.value, w.value), np.uint8)
            der = np.empty((h.value, w.value, 2), np.int16)
            lib().orc_pyr_get_level(self._pyr, l, _p(img), _p(der))
        return img, der

    def detect_keypoints(self, existing, max_points, min_dist):
        existing = np.ascontiguousarray(existing, dtype=np.float64).reshape(-1, 2)
        out = np.empty((max(max_points, 1), 2), np.float64)
        n = lib().orc_detect_keypoints(_p(self.image), self.w, self.h, self.w, _p(existing), len(existing),
                                       int(max_points), float(min_dist), _p(out))
        return np.concatenate([existing, out[:n]], axis=0)

    def track_keypoints(self, nxt, curr, guess=None, stats=None):
        curr = np.ascontiguousarray(curr, dtype=np.float64).reshape(-1, 2)
        n = len(curr)
        if guess is None:
            nx = np.zeros_like(curr)
            has = 0
        else:
            nx = np.ascontiguousarray(guess, dtype=np.float64).reshape(-1, 2).copy()
            has = 1
        status = np.zeros(n, np.uint8)
        st = stats if stats is not None else LkStats()
        lib().orc_track_keypoints(self._pyr, nxt._pyr, _p(curr), _p(nx), has, _p(status), n, C.byref(st))
        return nx, status

    def lk(self, nxt, prev_pts, next_pts, max_level=3, max_count=30, eps=0.01, win=21):
        prev_pts = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
        nx = np.ascontiguousarray(next_pts, dtype=np.float32).reshape(-1, 2).copy()
        status = np.zeros(len(prev_pts), np.uint8)
        st = LkStats()
        lib().orc_lk(self._pyr, nxt._pyr, _p(prev_pts), _p(nx), _p(status), len(prev_pts), win, max_level, max_count,
                     float(eps), C.byref(st))
        return nx, status, st


def clahe(gray, clip=6.0, tx=8, ty=8):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.empty_like(gray)
    lib().orc_clahe(_p(gray), w, h, w, float(clip), tx, ty, _p(out), w)
    return out


def harris_response(gray, k=0.04):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.float32)
    lib().orc_harris_response(_p(gray), w, h, w, float(k), _p(out))
    return out


def gftt(gray, max_corners, quality=1e-3, min_distance=20.0, k=0.04):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    cap = max_corners if max_corners > 0 else w * h
    xy = np.empty((cap, 2), np.float32)
    q = np.empty(cap, np.float32)
    n = lib().orc_gftt(_p(gray), w, h, w, int(max_corners), float(quality), float(min_distance), float(k), _p(xy),
                       _p(q))
    return xy[:n].copy(), q[:n].copy()


def poisson_select(pts, radius):
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    keep = np.zeros(len(pts), np.uint8)
    if len(pts):
        lib().orc_poisson_select(_p(pts), len(pts), float(radius), _p(keep))
    return keep


def set_threads(n):
    """OpenMP threads of the row / tile / point loops (default 1); results do not depend on it."""
    lib().orc_set_threads(int(n))


def get_threads():
    return lib().orc_get_threads()
