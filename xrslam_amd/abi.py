"""ctypes mirrors of the plain-C structs declared in include/xrslam_hip.h (layout only, no logic)."""
import copy
import ctypes as C

import numpy as np

STATE_DIM = 16
ES_DIM = 15
IMU_DIM = 281
FIX_POSE = 1
FIX_MOTION = 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_uint8)


class BaProblem(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int), ("frame_state", _dp), ("frame_fix", _bp),
        ("cam_q_bc", C.c_double * 4), ("cam_p_bc", C.c_double * 3),
        ("imu_q_bi", C.c_double * 4), ("imu_p_bi", C.c_double * 3),
        ("sqrt_inv_cov", C.c_double * 2),
        ("n_landmarks", C.c_int), ("inv_depth", _dp), ("landmark_fix", _bp),
        ("n_obs", C.c_int), ("obs_tgt", _ip), ("obs_ref", _ip), ("obs_lm", _ip),
        ("obs_z_tgt", _dp), ("obs_z_ref", _dp),
        ("n_rot", C.c_int), ("rot_tgt", _ip), ("rot_ref", _ip), ("rot_z_tgt", _dp), ("rot_z_ref", _dp),
        ("n_imu", C.c_int), ("imu_i", _ip), ("imu_j", _ip), ("imu_data", _dp),
        ("prior_n", C.c_int), ("prior_frames", _ip), ("prior_sqrt_info", _dp), ("prior_infovec", _dp),
        ("prior_lin", _dp),
        ("max_iterations", C.c_int),
    ]


class BaSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("termination", C.c_int),
                ("usable", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("ms_solve", C.c_double)]


class MargProblem(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int), ("victim", C.c_int), ("frame_state", _dp),
        ("cam_q_bc", C.c_double * 4), ("cam_p_bc", C.c_double * 3),
        ("imu_q_bi", C.c_double * 4), ("imu_p_bi", C.c_double * 3),
        ("sqrt_inv_cov", C.c_double * 2),
        ("prior_n", C.c_int), ("prior_frames", _ip), ("prior_sqrt_info", _dp), ("prior_infovec", _dp),
        ("prior_lin", _dp),
        ("n_imu", C.c_int), ("imu_i", _ip), ("imu_j", _ip), ("imu_data", _dp),
        ("n_landmarks", C.c_int), ("inv_depth", _dp),
        ("n_obs", C.c_int), ("obs_tgt", _ip), ("obs_ref", _ip), ("obs_lm", _ip),
        ("obs_z_tgt", _dp), ("obs_z_ref", _dp),
    ]


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _dpp(a):
    return a.ctypes.data_as(_dp)


def _ipp(a):
    return a.ctypes.data_as(_ip)


class _Factors:
    """shared unpacking of the factor dictionaries"""

    def _set_obs(self, o):
        o = o or {}
        self.obs_tgt = _arr(o.get("tgt", []), np.int32)
        self.obs_ref = _arr(o.get("ref", []), np.int32)
        self.obs_lm = _arr(o.get("lm", []), np.int32)
        self.obs_z_tgt = _arr(o.get("z_tgt", np.zeros((0, 3))), np.float64).reshape(-1, 3)
        self.obs_z_ref = _arr(o.get("z_ref", np.zeros((0, 3))), np.float64).reshape(-1, 3)

    def _set_imu(self, m):
        m = m or {}
        self.imu_i = _arr(m.get("i", []), np.int32)
        self.imu_j = _arr(m.get("j", []), np.int32)
        self.imu_data = _arr(m.get("data", np.zeros((0, IMU_DIM))), np.float64).reshape(-1, IMU_DIM)

    def _set_prior(self, p):
        p = p or {}
        self.prior_frames = _arr(p.get("frames", []), np.int32)
        n = ES_DIM * len(self.prior_frames)
        self.prior_sqrt_info = _arr(p.get("sqrt_info", np.zeros((n, n))), np.float64).reshape(n, n)
        self.prior_infovec = _arr(p.get("infovec", np.zeros(n)), np.float64)
        self.prior_lin = _arr(p.get("lin", np.zeros((len(self.prior_frames), STATE_DIM))), np.float64)

    def _fill_common(self, s):
        s.cam_q_bc[:] = list(self.cam_ext[:4])
        s.cam_p_bc[:] = list(self.cam_ext[4:7])
        s.imu_q_bi[:] = list(self.imu_ext[:4])
        s.imu_p_bi[:] = list(self.imu_ext[4:7])
        s.sqrt_inv_cov[:] = list(self.sqrt_inv_cov[:2])
        s.n_obs = len(self.obs_tgt)
        s.obs_tgt, s.obs_ref, s.obs_lm = _ipp(self.obs_tgt), _ipp(self.obs_ref), _ipp(self.obs_lm)
        s.obs_z_tgt, s.obs_z_ref = _dpp(self.obs_z_tgt), _dpp(self.obs_z_ref)
        s.n_imu = len(self.imu_i)
        s.imu_i, s.imu_j, s.imu_data = _ipp(self.imu_i), _ipp(self.imu_j), _dpp(self.imu_data)
        s.prior_n = len(self.prior_frames)
        s.prior_frames = _ipp(self.prior_frames)
        s.prior_sqrt_info, s.prior_infovec = _dpp(self.prior_sqrt_info), _dpp(self.prior_infovec)
        s.prior_lin = _dpp(self.prior_lin)


class BaProblemData(_Factors):
    """numpy-side owner of a BA problem; .struct() returns the C view (arrays stay alive with self)."""

    def __init__(self, frame_state, frame_fix, cam_ext, imu_ext, sqrt_inv_cov, inv_depth, landmark_fix=None,
                 obs=None, rot=None, imu=None, prior=None, max_iterations=30):
        self.frame_state = _arr(frame_state, np.float64).reshape(-1, STATE_DIM).copy()
        self.frame_fix = _arr(frame_fix, np.uint8).copy()
        self.cam_ext = _arr(cam_ext, np.float64)
        self.imu_ext = _arr(imu_ext, np.float64)
        self.sqrt_inv_cov = _arr(sqrt_inv_cov, np.float64)
        self.inv_depth = _arr(inv_depth, np.float64).copy()
        nl = len(self.inv_depth)
        self.landmark_fix = _arr(landmark_fix if landmark_fix is not None else np.zeros(nl), np.uint8).copy()
        self._set_obs(obs)
        r = rot or {}
        self.rot_tgt = _arr(r.get("tgt", []), np.int32)
        self.rot_ref = _arr(r.get("ref", []), np.int32)
        self.rot_z_tgt = _arr(r.get("z_tgt", np.zeros((0, 3))), np.float64).reshape(-1, 3)
        self.rot_z_ref = _arr(r.get("z_ref", np.zeros((0, 3))), np.float64).reshape(-1, 3)
        self._set_imu(imu)
        self._set_prior(prior)
        self.max_iterations = int(max_iterations)

    def copy(self):
        return copy.deepcopy(self)

    def struct(self):
        s = BaProblem()
        s.n_frames = len(self.frame_state)
        s.frame_state = _dpp(self.frame_state)
        s.frame_fix = self.frame_fix.ctypes.data_as(_bp)
        s.n_landmarks = len(self.inv_depth)
        s.inv_depth = _dpp(self.inv_depth)
        s.landmark_fix = self.landmark_fix.ctypes.data_as(_bp)
        self._fill_common(s)
        s.n_rot = len(self.rot_tgt)
        s.rot_tgt, s.rot_ref = _ipp(self.rot_tgt), _ipp(self.rot_ref)
        s.rot_z_tgt, s.rot_z_ref = _dpp(self.rot_z_tgt), _dpp(self.rot_z_ref)
        s.max_iterations = self.max_iterations
        return s


class MargProblemData(_Factors):
    def __init__(self, frame_state, victim, cam_ext, imu_ext, sqrt_inv_cov, prior, imu, inv_depth, obs):
        self.frame_state = _arr(frame_state, np.float64).reshape(-1, STATE_DIM).copy()
        self.victim = int(victim)
        self.cam_ext = _arr(cam_ext, np.float64)
        self.imu_ext = _arr(imu_ext, np.float64)
        self.sqrt_inv_cov = _arr(sqrt_inv_cov, np.float64)
        self._set_prior(prior)
        self._set_imu(imu)
        self.inv_depth = _arr(inv_depth, np.float64).copy()
        self._set_obs(obs)

    def struct(self):
        s = MargProblem()
        s.n_frames = len(self.frame_state)
        s.victim = self.victim
        s.frame_state = _dpp(self.frame_state)
        self._fill_common(s)
        s.n_landmarks = len(self.inv_depth)
        s.inv_depth = _dpp(self.inv_depth)
        return s
