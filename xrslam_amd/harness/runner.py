"""The player's call sequence (reference xrslam-pc/player/src/main.cpp:116-169) over the outer C ABI
(include/XRSLAM.h): at equal timestamps the asynchronous dataset reader yields gyroscope, then
accelerometer, then camera (IO/async_dataset_reader.cpp:41-48).  Works against any shared object that
exports the XRSLAM.h symbols (the product library, or the oracle-backed CPU reference build)."""
import ctypes as C
import os

import numpy as np

XRSLAM_SENSOR_CAMERA, XRSLAM_SENSOR_ACCELERATION, XRSLAM_SENSOR_GYROSCOPE = 0, 2, 3
XRSLAM_RESULT_BODY_POSE, XRSLAM_RESULT_STATE = 0, 2


class XRSLAMImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("timeStamp", C.c_double), ("stride", C.c_int), ("camera_id", C.c_int),
                ("channel", C.c_int), ("ext", C.c_void_p)]


class XRSLAMVec3(C.Structure):       # XRSLAMAcceleration / XRSLAMGyroscope
    _fields_ = [("data", C.c_double * 3), ("timestamp", C.c_double)]


class XRSLAMPose(C.Structure):
    _fields_ = [("quaternion", C.c_double * 4), ("translation", C.c_double * 3), ("timestamp", C.c_double)]


class BaStats(C.Structure):   # xrhip_ba_stats (include/xrslam_hip.h)
    _fields_ = [("n_solve_try", C.c_long), ("n_trials", C.c_long), ("ms_solve_try", C.c_double), ("n_timed", C.c_long),
                ("flops_solve_try", C.c_double), ("n_tiny", C.c_long), ("n_chain_timed", C.c_long), ("ms_chain", C.c_double),
                ("bytes_chain", C.c_double)]


class XRSLAMAmdTimes(C.Structure):
    _fields_ = [("frames", C.c_long), ("solves", C.c_long), ("solve_iterations", C.c_long),
                ("marginalizations", C.c_long), ("keyframes", C.c_long), ("ba_device_ms", C.c_double),
                ("wall_preprocess", C.c_double), ("wall_track", C.c_double), ("wall_detect", C.c_double),
                ("wall_preintegrate", C.c_double), ("wall_solve", C.c_double), ("wall_marginalize", C.c_double),
                ("wall_frame", C.c_double), ("wall_scope", C.c_double * 16)]


class GroupStats(C.Structure):   # xrhip_group_stats (include/xrslam_hip.h)
    _fields_ = [("batches", C.c_longlong * 12), ("entries", C.c_longlong * 12), ("ms", C.c_double * 12), ("timed", C.c_longlong * 12)]


GROUP_KINDS = ("call", "upload", "preprocess", "track", "detect", "chain", "preint", "window_round", "window_trials", None, None, "gate")   # gate: openings / all present / timeouts


class XRSLAMAmdInitReport(C.Structure):
    _fields_ = [("attempts", C.c_long), ("successes", C.c_long), ("sfm_candidate", C.c_int),
                ("sfm_triangulated", C.c_int), ("scale", C.c_double), ("gravity", C.c_double * 3),
                ("bg", C.c_double * 3)]


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SLAM_YAML = os.path.join(ROOT, "configs", "euroc_slam.yaml")
SENSOR_YAML = os.path.join(ROOT, "configs", "euroc_sensor.yaml")


def load(lib_path):
    lib = C.CDLL(lib_path)
    lib.XRSLAMCreate.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.XRSLAMPushSensorData.argtypes = [C.c_int, C.c_void_p]
    lib.XRSLAMPushSensorData.restype = None
    lib.XRSLAMRunOneFrame.restype = None
    lib.XRSLAMGetResult.argtypes = [C.c_int, C.c_void_p]
    lib.XRSLAMGetResult.restype = None
    lib.XRSLAMDestroy.restype = None
    lib.XRSLAMAmdSetInitialState.argtypes = [C.c_double] + [C.c_void_p] * 5
    lib.XRSLAMAmdSetInitialState.restype = None
    lib.XRSLAMAmdPushImageDevice.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.XRSLAMAmdPushImageDevice.restype = None
    lib.XRSLAMAmdGetTimes.argtypes = [C.POINTER(XRSLAMAmdTimes)]
    lib.XRSLAMAmdGetTimes.restype = None
    lib.XRSLAMAmdLastError.restype = C.c_char_p
    lib.XRSLAMAmdGetInitReport.argtypes = [C.POINTER(XRSLAMAmdInitReport)]
    lib.XRSLAMAmdGetInitReport.restype = None
    lib.XRSLAMAmdSetProfiling.argtypes = [C.c_int]
    lib.XRSLAMAmdSetProfiling.restype = None
    lib.XRSLAMAmdGetKltStats.argtypes = [C.c_void_p, C.c_int]
    lib.XRSLAMAmdGetKltStats.restype = None
    lib.XRSLAMAmdGetBaStats.argtypes = [C.c_void_p, C.c_int]
    lib.XRSLAMAmdGetBaStats.restype = None
    if hasattr(lib, "XRSLAMAmdSetDeviceUndistort"):
        lib.XRSLAMAmdSetDeviceUndistort.argtypes = [C.c_char_p]
        lib.XRSLAMAmdSetDeviceUndistort.restype = None
    lib.XRSLAMAmdSetThreading.argtypes = [C.c_int]
    lib.XRSLAMAmdSetThreading.restype = None
    lib.XRSLAMAmdFlush.restype = None
    if hasattr(lib, "XRSLAMAmdInstanceCreate"):   # instance-scoped forms: the instance handle is the first argument
        H = C.c_void_p
        lib.XRSLAMAmdInstanceCreate.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(H), C.POINTER(C.c_void_p)]
        for name, args, res in (("Destroy", [], None), ("PushSensorData", [C.c_int, C.c_void_p], None),
                                ("RunOneFrame", [], None), ("GetResult", [C.c_int, C.c_void_p], None),
                                ("SetInitialState", [C.c_double] + [C.c_void_p] * 5, None),
                                ("PushImageDevice", [C.c_void_p, C.c_int, C.c_double], None),
                                ("GetTimes", [C.POINTER(XRSLAMAmdTimes)], None), ("SetProfiling", [C.c_int], None),
                                ("GetBaStats", [C.c_void_p, C.c_int], None), ("GetKltStats", [C.c_void_p, C.c_int], None),
                                ("GetInitReport", [C.POINTER(XRSLAMAmdInitReport)], None), ("LastError", [], C.c_char_p),
                                ("SetDeviceUndistort", [C.c_char_p], None), ("SetThreading", [C.c_int], None),
                                ("Flush", [], None)):
            fn = getattr(lib, "XRSLAMAmdInstance" + name)
            fn.argtypes = [H] + args
            fn.restype = res
        lib.XRSLAMAmdInstanceReplay.argtypes = [H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p]
        lib.XRSLAMAmdInstanceReplay.restype = C.c_int
    if hasattr(lib, "XRSLAMAmdGroupCreate"):
        lib.XRSLAMAmdGroupCreate.argtypes = [C.POINTER(C.c_void_p)]
        lib.XRSLAMAmdGroupDestroy.argtypes = [C.c_void_p]
        lib.XRSLAMAmdInstanceJoinGroup.argtypes = [C.c_void_p, C.c_void_p]
        lib.XRSLAMAmdGroupSetProfiling.argtypes = [C.c_void_p, C.c_int]
        lib.XRSLAMAmdGroupSetProfiling.restype = None
        lib.XRSLAMAmdGroupGetStats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.XRSLAMAmdGroupGetStats.restype = None
    return lib


class Group:
    """An instance group (XRSLAMAmdGroup, include/XRSLAM.h): the sessions created with group=<this> share their per-frame launches."""

    def __init__(self, lib_path):
        self.lib = load(lib_path)
        self.handle = C.c_void_p()
        if self.lib.XRSLAMAmdGroupCreate(C.byref(self.handle)) != 1:
            raise RuntimeError("XRSLAMAmdGroupCreate failed: %s" % self.lib.XRSLAMAmdLastError().decode())

    def set_profiling(self, on):
        self.lib.XRSLAMAmdGroupSetProfiling(self.handle, 1 if on else 0)

    def stats(self, reset=False):
        """-> {kind: {"batches", "requests", "ms", "timed"}} for the kinds that saw traffic"""
        st = GroupStats()
        self.lib.XRSLAMAmdGroupGetStats(self.handle, C.byref(st), 1 if reset else 0)
        return {name: {"batches": int(st.batches[i]), "requests": int(st.entries[i]), "ms": float(st.ms[i]), "timed": int(st.timed[i])}
                for i, name in enumerate(GROUP_KINDS) if name and st.batches[i]}

    def close(self):
        if self.handle:
            if self.lib.XRSLAMAmdGroupDestroy(self.handle) != 1:
                raise RuntimeError("XRSLAMAmdGroupDestroy failed: %s" % self.lib.XRSLAMAmdLastError().decode())
            self.handle = None


class _Api:
    """The entry points a Session drives: the process-global ones (the reference's six symbols + XRSLAMAmd*), or their
    XRSLAMAmdInstance* forms bound to one instance handle."""

    def __init__(self, lib, handle=None):
        import functools
        names = {"push": ("XRSLAMPushSensorData", "XRSLAMAmdInstancePushSensorData"),
                 "run": ("XRSLAMRunOneFrame", "XRSLAMAmdInstanceRunOneFrame"),
                 "get_result": ("XRSLAMGetResult", "XRSLAMAmdInstanceGetResult"),
                 "destroy": ("XRSLAMDestroy", "XRSLAMAmdInstanceDestroy"),
                 "set_initial_state": ("XRSLAMAmdSetInitialState", "XRSLAMAmdInstanceSetInitialState"),
                 "push_image_device": ("XRSLAMAmdPushImageDevice", "XRSLAMAmdInstancePushImageDevice"),
                 "get_times": ("XRSLAMAmdGetTimes", "XRSLAMAmdInstanceGetTimes"),
                 "set_profiling": ("XRSLAMAmdSetProfiling", "XRSLAMAmdInstanceSetProfiling"),
                 "get_ba_stats": ("XRSLAMAmdGetBaStats", "XRSLAMAmdInstanceGetBaStats"),
                 "get_klt_stats": ("XRSLAMAmdGetKltStats", "XRSLAMAmdInstanceGetKltStats"),
                 "get_init_report": ("XRSLAMAmdGetInitReport", "XRSLAMAmdInstanceGetInitReport"),
                 "last_error": ("XRSLAMAmdLastError", "XRSLAMAmdInstanceLastError"),
                 "set_device_undistort": ("XRSLAMAmdSetDeviceUndistort", "XRSLAMAmdInstanceSetDeviceUndistort"),
                 "set_threading": ("XRSLAMAmdSetThreading", "XRSLAMAmdInstanceSetThreading"),
                 "sync": ("XRSLAMAmdFlush", "XRSLAMAmdInstanceFlush")}
        for attr, (glob, inst) in names.items():
            setattr(self, attr, getattr(lib, glob) if handle is None else functools.partial(getattr(lib, inst), handle))


class Session:
    """One XRSLAM instance: the process singleton behind the reference's six symbols (XRSLAMManager.cpp:6-9), or --
    instance=True -- an XRSLAMAmdInstance of its own, so that several sessions can live in one process."""

    def __init__(self, lib_path, seq, slam_yaml=SLAM_YAML, sensor_yaml=SENSOR_YAML, device_frames=None,
                 init_frames=60, instance=False, device_undistort=None, threading=0, group=None):
        self.lib = load(lib_path)
        self.seq = seq
        cfg = C.c_void_p()
        if instance:
            handle = C.c_void_p()
            ok = self.lib.XRSLAMAmdInstanceCreate(slam_yaml.encode(), sensor_yaml.encode(), C.byref(handle), C.byref(cfg))
            if ok != 1:
                raise RuntimeError("XRSLAMAmdInstanceCreate failed: %s" % self.lib.XRSLAMAmdLastError().decode())
            self.api = _Api(self.lib, handle)
            self._handle = handle
            if group is not None and self.lib.XRSLAMAmdInstanceJoinGroup(handle, group.handle) != 1:
                raise RuntimeError("XRSLAMAmdInstanceJoinGroup failed: %s" % self.api.last_error().decode())
        else:
            ok = self.lib.XRSLAMCreate(slam_yaml.encode(), sensor_yaml.encode(), b"", b"xrslam_amd", C.byref(cfg))
            if ok != 1:
                raise RuntimeError("XRSLAMCreate failed: %s" % self.lib.XRSLAMAmdLastError().decode())
            self.api = _Api(self.lib)
        if threading:   # 1 = backend of frame t beside the feature tracker of frame t+1 (XRSLAMAmdSetThreading)
            self.api.set_threading(int(threading))
        if device_undistort:   # frames are pushed as the camera recorded them and rectified on the GPU
            self.api.set_device_undistort(device_undistort.encode())
        st = seq["states"]
        for i in range(min(init_frames, len(st))):
            s = np.ascontiguousarray(st[i])
            q, p, v, bg, ba = [np.ascontiguousarray(s[a:b]) for a, b in ((0, 4), (4, 7), (7, 10), (10, 13), (13, 16))]
            self.api.set_initial_state(float(seq["cam_t"][i]), q.ctypes.data, p.ctypes.data, v.ctypes.data,
                                       bg.ctypes.data, ba.ctypes.data)
        self.device_frames = device_frames   # (base pointer, bytes per frame, stride) when frames live in HBM
        self.imu_k = 0
        self.frame_k = 0
        self.poses = []

    def _imu_structs(self):
        """The sensor structs of the whole sequence, built once (the player reads them from a file the same way);
        pushing a sample is then a single foreign call."""
        imu = self.seq["imu"]
        n = len(imu)
        gy, ac = (XRSLAMVec3 * n)(), (XRSLAMVec3 * n)()
        for k in range(n):
            r = imu[k]
            gy[k].data[0], gy[k].data[1], gy[k].data[2], gy[k].timestamp = r[1], r[2], r[3], r[0]
            ac[k].data[0], ac[k].data[1], ac[k].data[2], ac[k].timestamp = r[4], r[5], r[6], r[0]
        self._gy, self._ac = gy, ac
        self._imu_t = [float(v) for v in imu[:, 0]]

    def _push_imu_until(self, t_limit):
        if not hasattr(self, "_gy"):
            self._imu_structs()
        push, gy, ac, ts, n = self.api.push, self._gy, self._ac, self._imu_t, len(self._imu_t)
        k = self.imu_k
        lim = t_limit + 1e-9
        while k < n and ts[k] <= lim:
            push(XRSLAM_SENSOR_GYROSCOPE, C.byref(gy[k]))
            push(XRSLAM_SENSOR_ACCELERATION, C.byref(ac[k]))
            k += 1
        self.imu_k = k

    def step(self):
        """Feeds everything up to and including the next camera frame; returns False at the end."""
        if self.frame_k >= len(self.seq["cam_t"]):
            return False
        t = float(self.seq["cam_t"][self.frame_k])
        self._push_imu_until(t)
        if self.device_frames is not None:
            base, fbytes, stride = self.device_frames
            self.api.push_image_device(C.c_void_p(base + self.frame_k * fbytes), stride, t)
        else:
            fr = self.seq["frames"][self.frame_k]
            img = XRSLAMImage(fr.ctypes.data, t, fr.strides[0], 0, 1, None)
            self.api.push(XRSLAM_SENSOR_CAMERA, C.byref(img))
        self.api.run()
        state = C.c_int(-1)
        self.api.get_result(XRSLAM_RESULT_STATE, C.byref(state))
        if state.value == 1:
            pose = XRSLAMPose()
            self.api.get_result(XRSLAM_RESULT_BODY_POSE, C.byref(pose))
            self.poses.append([pose.timestamp] + list(pose.translation) + list(pose.quaternion))
        self.frame_k += 1
        return True

    def step_n(self, n):
        """n camera frames through XRSLAMAmdInstanceReplay: the same call sequence as n x step(), issued natively (one
        foreign call, the interpreter lock released throughout) -- instance sessions only."""
        if not hasattr(self, "_rp"):
            imu = np.ascontiguousarray(self.seq["imu"], np.float64)
            cam_t = np.ascontiguousarray(self.seq["cam_t"], np.float64)
            frames = np.ascontiguousarray(self.seq["frames"])
            self._rp = (imu, cam_t, frames, C.c_int(self.imu_k), C.c_int(self.frame_k))
        imu, cam_t, frames, ic, fc = self._rp
        ic.value, fc.value = self.imu_k, self.frame_k
        out = np.zeros((max(n, 1), 8))
        if self.device_frames is not None:
            base, fbytes, stride = self.device_frames
            ptr, on_dev = C.c_void_p(base), 1
        else:
            ptr, fbytes, stride, on_dev = C.c_void_p(frames.ctypes.data), frames.strides[0], frames.strides[1], 0
        k = self.lib.XRSLAMAmdInstanceReplay(self._handle, imu.ctypes.data, len(imu), cam_t.ctypes.data, len(cam_t), ptr, fbytes,
                                             stride, on_dev, C.byref(ic), C.byref(fc), int(n), out.ctypes.data)
        if k < 0:
            raise RuntimeError("XRSLAMAmdInstanceReplay: bad arguments")
        self.imu_k, self.frame_k = ic.value, fc.value
        for r in out[:k]:
            self.poses.append([r[0]] + list(r[1:4]) + list(r[4:8]))
        return k

    def flush(self):
        """Pushes the IMU samples after the last frame so the last queued frame is processed
        (processing is triggered by the first IMU sample later than the frame, detail.cpp:130-142)."""
        self._push_imu_until(1e300)

    def sync(self):
        """Pipelined mode: waits for the backend job in flight (no-op otherwise)."""
        self.api.sync()

    def times(self):
        t = XRSLAMAmdTimes()
        self.api.get_times(C.byref(t))
        return t

    def set_profiling(self, on):
        self.api.set_profiling(1 if on else 0)

    def klt_stats(self, reset=False):
        from xrslam_amd.klt import KltStats
        st = KltStats()
        self.api.get_klt_stats(C.byref(st), 1 if reset else 0)
        return st

    def ba_stats(self, reset=False):
        st = BaStats()
        self.api.get_ba_stats(C.byref(st), 1 if reset else 0)
        return st

    def init_report(self):
        r = XRSLAMAmdInitReport()
        self.api.get_init_report(C.byref(r))
        return r

    def error(self):
        return self.api.last_error().decode()

    def close(self):
        self.api.destroy()


def ate_rmse(poses, seq):
    """ATE RMSE after SE(3) Umeyama alignment (what `evo_ape tum -a` computes, docs/en/tutorials/euroc_evaluation.md)."""
    if len(poses) < 3:
        return float("nan")
    P = np.array(poses)
    P = P[np.abs(P[:, 4:8]).sum(1) > 0]      # results before the first tracked frame are the all-zero pose (detail.cpp:165-168)
    if len(P) < 3:
        return float("nan")
    idx = np.searchsorted(seq["cam_t"], P[:, 0] - 1e-6)
    idx = np.clip(idx, 0, len(seq["cam_t"]) - 1)
    gt = seq["states"][idx, 4:7]
    est = P[:, 1:4]
    mu_e, mu_g = est.mean(0), gt.mean(0)
    H = (est - mu_e).T @ (gt - mu_g)
    U, S, Vt = np.linalg.svd(H)
    D = np.eye(3)
    if np.linalg.det(Vt.T @ U.T) < 0:
        D[2, 2] = -1
    R = Vt.T @ D @ U.T
    aligned = (R @ (est - mu_e).T).T + mu_g
    return float(np.sqrt(((aligned - gt) ** 2).sum(1).mean()))
