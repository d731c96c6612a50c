"""ctypes mirror of plug point #1 (xrslam::Image, xrslam/include/xrslam/xrslam.h:137-161)
on top of the C ABI in include/xrslam_hip.h.  Method names and argument meaning
follow xrslam::extra::OpenCvImage (xrslam-extra/src/xrslam/extra/opencv_image.cpp)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class KltStats(C.Structure):
    _fields_ = [("ms_preprocess", C.c_double), ("ms_track", C.c_double), ("ms_detect", C.c_double),
                ("n_preprocess", C.c_longlong), ("n_track", C.c_longlong), ("n_detect", C.c_longlong),
                ("lk_templates", C.c_longlong), ("lk_iterations", C.c_longlong), ("lk_points", C.c_longlong),
                ("detect_full_list", C.c_longlong)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bind():
    L = _lib.lib()
    vp = C.c_void_p
    L.xrhip_klt_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.xrhip_klt_destroy.argtypes = [vp]
    L.xrhip_klt_destroy.restype = None
    L.xrhip_image_create.argtypes = [vp, C.POINTER(vp)]
    L.xrhip_image_upload.argtypes = [vp, vp, C.c_int]
    L.xrhip_image_upload_device.argtypes = [vp, vp, C.c_int]
    L.xrhip_image_destroy.argtypes = [vp]
    L.xrhip_image_destroy.restype = None
    L.xrhip_image_preprocess.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    L.xrhip_image_release.argtypes = [vp]
    L.xrhip_image_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, vp, C.POINTER(C.c_int)]
    L.xrhip_image_track.argtypes = [vp, vp, vp, vp, C.c_int, vp, C.c_int]
    L.xrhip_image_lk.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.xrhip_image_level_dims.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.xrhip_image_download_level.argtypes = [vp, C.c_int, vp, vp]
    L.xrhip_image_download_harris.argtypes = [vp, vp]
    L.xrhip_klt_set_profiling.argtypes = [vp, C.c_int]
    L.xrhip_klt_get_stats.argtypes = [vp, C.POINTER(KltStats), C.c_int]
    L.xrhip_klt_synchronize.argtypes = [vp]
    L.xrhip_klt_set_undistort_map.argtypes = [vp, vp]
    L.xrhip_image_upload_distorted.argtypes = [vp, vp, C.c_int, C.c_int]
    L.xrhip_debug_get_raw.argtypes = [vp, vp]
    L.xrhip_debug_set_fused_pyramid.argtypes = [vp, C.c_int]
    L.xrhip_debug_get_level_padded.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


_L = None


def L():
    global _L
    if _L is None:
        _L = _bind()
    return _L


class KltContext:
    """One per sequence (owns the HIP stream and the CLAHE/GFTT state the reference keeps in statics)."""

    def __init__(self, width, height, max_points=200):
        self.w, self.h = int(width), int(height)
        h = C.c_void_p()
        check(L().xrhip_klt_create(self.w, self.h, int(max_points), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            L().xrhip_klt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def image(self, gray=None):
        return HipImage(self, gray)

    def set_profiling(self, on):
        check(L().xrhip_klt_set_profiling(self._h, 1 if on else 0))

    def stats(self, reset=False):
        s = KltStats()
        check(L().xrhip_klt_get_stats(self._h, C.byref(s), 1 if reset else 0))
        return s

    def synchronize(self):
        check(L().xrhip_klt_synchronize(self._h))

    def set_fused_pyramid(self, on):
        """Development / parity switch: preprocess() builds the pyramid in one launch (default) or in the five it replaces."""
        check(L().xrhip_debug_set_fused_pyramid(self._h, 1 if on else 0))

    def set_undistort_map(self, map2):
        """Packed 1/32-pixel inverse map [h][w][2] uint32 (include/xrslam_hip.h), or None to switch the device
        undistortion off."""
        if map2 is None:
            check(L().xrhip_klt_set_undistort_map(self._h, None))
            return
        map2 = np.ascontiguousarray(map2, dtype=np.uint32)
        assert map2.shape == (self.h, self.w, 2), map2.shape
        check(L().xrhip_klt_set_undistort_map(self._h, _p(map2)))


class HipImage:
    """xrslam::Image on the MI355X."""

    def __init__(self, ctx, gray=None):
        self.ctx = ctx
        h = C.c_void_p()
        check(L().xrhip_image_create(ctx._h, C.byref(h)))
        self._h = h
        self.t = 0.0
        if gray is not None:
            self.upload(gray)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            L().xrhip_image_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def width(self):
        return self.ctx.w

    def height(self):
        return self.ctx.h

    def level_num(self):
        return 3

    def upload(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        assert gray.shape == (self.ctx.h, self.ctx.w), gray.shape
        check(L().xrhip_image_upload(self._h, _p(gray), gray.strides[0]))

    def upload_distorted(self, gray):
        """The frame as the camera recorded it: rectified on the device (KltContext.set_undistort_map)."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        assert gray.shape == (self.ctx.h, self.ctx.w), gray.shape
        check(L().xrhip_image_upload_distorted(self._h, _p(gray), gray.strides[0], 0))

    def upload_distorted_device(self, dev_ptr, stride):
        check(L().xrhip_image_upload_distorted(self._h, C.c_void_p(dev_ptr), int(stride), 1))

    def raw(self):
        """The 8-bit frame preprocess() will read (parity aid)."""
        out = np.empty((self.ctx.h, self.ctx.w), np.uint8)
        check(L().xrhip_debug_get_raw(self._h, _p(out)))
        return out

    def upload_device(self, dev_ptr, stride):
        check(L().xrhip_image_upload_device(self._h, C.c_void_p(int(dev_ptr)), int(stride)))

    def preprocess(self, clip=6.0, tiles_x=8, tiles_y=8):
        check(L().xrhip_image_preprocess(self._h, float(clip), int(tiles_x), int(tiles_y)))

    def release_image_buffer(self):
        check(L().xrhip_image_release(self._h))

    def prefetch_detect(self):
        """Hint: detect_keypoints will follow; the Harris pass is queued behind the next tracking launch onto this image."""
        check(L().xrhip_image_prefetch_detect(self._h))

    def detect_keypoints(self, existing, max_points, min_dist):
        existing = np.ascontiguousarray(existing, dtype=np.float64).reshape(-1, 2)
        out = np.empty((max(int(max_points), 1), 2), np.float64)
        n = C.c_int(0)
        check(L().xrhip_image_detect(self._h, _p(existing), len(existing), int(max_points), float(min_dist), _p(out),
                                     C.byref(n)))
        return np.concatenate([existing, out[:n.value]], axis=0)

    def track_keypoints(self, nxt, curr, guess=None):
        curr = np.ascontiguousarray(curr, dtype=np.float64).reshape(-1, 2)
        n = len(curr)
        if guess is None:
            nx = np.zeros_like(curr)
            has = 0
        else:
            nx = np.ascontiguousarray(guess, dtype=np.float64).reshape(-1, 2).copy()
            has = 1
        status = np.zeros(n, np.uint8)
        check(L().xrhip_image_track(self._h, nxt._h, _p(curr), _p(nx), has, _p(status), n))
        return nx, status

    def lk(self, nxt, prev_pts, next_pts):
        prev_pts = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
        nx = np.ascontiguousarray(next_pts, dtype=np.float32).reshape(-1, 2).copy()
        status = np.zeros(len(prev_pts), np.uint8)
        check(L().xrhip_image_lk(self._h, nxt._h, _p(prev_pts), _p(nx), _p(status), len(prev_pts)))
        return nx, status

    def level(self, l):
        w = C.c_int()
        h = C.c_int()
        check(L().xrhip_image_level_dims(self._h, l, C.byref(w), C.byref(h)))
        img = np.empty((h.value, w.value), np.uint8)
        der = np.empty((h.value, w.value, 2), np.int16)
        check(L().xrhip_image_download_level(self._h, l, _p(img), _p(der)))
        return img, der

    def level_padded(self, l):
        """Level l's image plane with its 21-pixel reflect-101 border (what the LK windows read near the edges)."""
        rows, cols = C.c_int(), C.c_int()
        check(L().xrhip_debug_get_level_padded(self._h, l, None, C.byref(rows), C.byref(cols)))
        out = np.empty((rows.value, cols.value), np.uint8)
        check(L().xrhip_debug_get_level_padded(self._h, l, _p(out), C.byref(rows), C.byref(cols)))
        return out

    def harris(self):
        out = np.empty((self.ctx.h, self.ctx.w), np.float32)
        check(L().xrhip_image_download_harris(self._h, _p(out)))
        return out
