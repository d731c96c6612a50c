"""Unit test of the PRODUCT's device math (xrslam_amd/csrc/ba_math.hip.h) compiled for the host with
hipcc (every function there is __host__ __device__), checked against the oracle.  This is the same
source the gfx950 kernels execute; it lets the factor arithmetic be verified without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import ba_oracle as bo
from tests import ba_synth as bs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_check", "ba_math_host.cpp")
OUT = os.path.join(ROOT, "tests", "host_check", "_build", "libba_math_host.so")


@pytest.fixture(scope="module")
def hc():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    hdr = os.path.join(ROOT, "xrslam_amd", "csrc", "ba_math.hip.h")
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC",
                               "-ffp-contract=off", "-shared", SRC, "-o", OUT])
    return C.CDLL(OUT)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_factor_math_matches_oracle(hc):
    pd, _ = bs.make_window(K=5, L=60, seed=13)
    st = pd.frame_state
    cam, sic = np.ascontiguousarray(bs.CAM_EXT), np.ascontiguousarray(bs.SQRT_INV_COV)
    for o in range(0, len(pd.obs_tgt), 5):
        ft, fr, l = pd.obs_tgt[o], pd.obs_ref[o], pd.obs_lm[o]
        zt, zr = np.ascontiguousarray(pd.obs_z_tgt[o]), np.ascontiguousarray(pd.obs_z_ref[o])
        r = np.zeros(2); Jt = np.zeros(12); Jr = np.zeros(12); Jl = np.zeros(2)
        hc.hc_reprojection(_p(st[ft].copy()), _p(st[fr].copy()), C.c_double(pd.inv_depth[l]), _p(zt), _p(zr), _p(cam),
                           _p(sic), _p(r), _p(Jt), _p(Jr), _p(Jl))
        ro, Jto, Jro, Jlo = bo.eval_reprojection(st[ft], st[fr], pd.inv_depth[l], zt, zr, cam, sic)
        np.testing.assert_allclose(r, ro, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(Jt.reshape(2, 6), Jto, rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(Jr.reshape(2, 6), Jro, rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(Jl, Jlo, rtol=1e-10, atol=1e-9)
        Jq = np.zeros(6)
        hc.hc_rotation(_p(st[ft].copy()), _p(st[fr].copy()), _p(zt), _p(zr), _p(cam), _p(sic), _p(r), _p(Jq))
        ro, Jqo = bo.eval_rotation(st[ft], st[fr], zt, zr, cam, sic)
        np.testing.assert_allclose(r, ro, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(Jq.reshape(2, 3), Jqo, rtol=1e-10, atol=1e-9)
    imu_ext = np.array([0.01, -0.02, 0.03, 0.9993, 0.02, -0.01, 0.03])
    imu_ext[:4] /= np.linalg.norm(imu_ext[:4])
    for ext in (np.ascontiguousarray(bs.IMU_EXT), imu_ext):
        for k in range(len(pd.imu_i)):
            i, j = pd.imu_i[k], pd.imu_j[k]
            ref = st[i, 10:16] + np.array([1e-4, -2e-4, 1e-4, 3e-3, -2e-3, 1e-3])
            r = np.zeros(15); Ji = np.zeros(225); Jj = np.zeros(225)
            hc.hc_imu(_p(st[i].copy()), _p(st[j].copy()), _p(pd.imu_data[k].copy()), _p(ref), _p(ext), _p(r), _p(Ji),
                      _p(Jj))
            ro, Jio, Jjo = bo.eval_imu(st[i], st[j], pd.imu_data[k], ref, ext)
            scale = np.abs(Jio).max()
            np.testing.assert_allclose(r, ro, rtol=1e-9, atol=1e-7)
            assert np.abs(Ji.reshape(15, 15) - Jio).max() < 1e-10 * scale
            assert np.abs(Jj.reshape(15, 15) - Jjo).max() < 1e-10 * scale


def test_imu_factor_in_pieces_is_the_single_lane_factor_bit_for_bit(hc):
    """kb_chain cuts an IMU factor's residual and Jacobians along their data dependencies and deals the pieces to its four
    wavefronts (ba_chain.hip.h).  Every piece uses the single-lane form's expression for its values, operand for operand: the
    pieces put together are the single-lane result, bit for bit (host compile of the same header; free and constant sides)."""
    pd, _ = bs.make_window(K=5, L=40, seed=21)
    st = pd.frame_state
    ext = np.array([0.01, -0.02, 0.03, 0.9993, 0.02, -0.01, 0.03])
    ext[:4] /= np.linalg.norm(ext[:4])
    for k in range(len(pd.imu_i)):
        i, j = pd.imu_i[k], pd.imu_j[k]
        ref = st[i, 10:16] + np.array([1e-4, -2e-4, 1e-4, 3e-3, -2e-3, 1e-3])
        for need_i, need_j in ((1, 1), (0, 1), (1, 0)):
            raw = np.zeros(30); Ji = np.zeros(450); Jj = np.zeros(450)
            hc.hc_imu_pieces(_p(st[i].copy()), _p(st[j].copy()), _p(pd.imu_data[k].copy()), _p(ref), _p(ext), need_i, need_j,
                             _p(raw), _p(Ji), _p(Jj))
            np.testing.assert_array_equal(raw[:15], raw[15:])
            np.testing.assert_array_equal(Ji[:225], Ji[225:])
            np.testing.assert_array_equal(Jj[:225], Jj[225:])
            assert np.abs(raw[:3]).max() > 0 and (not need_i or np.abs(Ji[:225]).max() > 0)


def test_constant_landmark_reprojection_is_the_general_factor_bit_for_bit(hc):
    """kb_chain evaluates reprojection factors against constant landmarks from per-solve constants (tangent basis, landmark in
    the reference rig / in the world): same expressions as eval_reprojection, hence the same bits for r, Jt and Jr."""
    pd, _ = bs.make_window(K=5, L=60, seed=14)
    st = pd.frame_state
    cam, sic = np.ascontiguousarray(bs.CAM_EXT), np.ascontiguousarray(bs.SQRT_INV_COV)
    for o in range(0, len(pd.obs_tgt), 3):
        ft, fr, l = pd.obs_tgt[o], pd.obs_ref[o], pd.obs_lm[o]
        zt, zr = np.ascontiguousarray(pd.obs_z_tgt[o]), np.ascontiguousarray(pd.obs_z_ref[o])
        for ref_free in (0, 1):
            r = np.zeros(4); Jt = np.zeros(24); Jr = np.zeros(24)
            hc.hc_reprojection_cached(_p(st[ft].copy()), _p(st[fr].copy()), C.c_double(pd.inv_depth[l]), _p(zt), _p(zr), _p(cam),
                                      _p(sic), ref_free, _p(r), _p(Jt), _p(Jr))
            np.testing.assert_array_equal(r[:2], r[2:])
            np.testing.assert_array_equal(Jt[:12], Jt[12:])
            np.testing.assert_array_equal(Jr[:12], Jr[12:])
            assert np.abs(Jt[:12]).max() > 0 and (not ref_free or np.abs(Jr[:12]).max() > 0)


def test_free_target_reprojection_from_frame_tables_is_the_general_factor_to_rounding(hc):
    """Round 6: kb_chain evaluates the factors of a free TARGET frame from the frame's R^T and p (once per frame) and the camera
    extrinsics' (once per solve) -- ~180 instead of ~450 instructions per factor, another order of operations: residual and
    Jacobian agree with eval_reprojection to rounding."""
    pd, _ = bs.make_window(K=5, L=60, seed=14)
    st = pd.frame_state
    cam, sic = np.ascontiguousarray(bs.CAM_EXT), np.ascontiguousarray(bs.SQRT_INV_COV)
    worst_r = worst_j = 0.0
    for o in range(0, len(pd.obs_tgt), 2):
        ft, fr, l = pd.obs_tgt[o], pd.obs_ref[o], pd.obs_lm[o]
        zt, zr = np.ascontiguousarray(pd.obs_z_tgt[o]), np.ascontiguousarray(pd.obs_z_ref[o])
        r = np.zeros(4); Jt = np.zeros(24)
        hc.hc_reprojection_tgt(_p(st[ft].copy()), _p(st[fr].copy()), C.c_double(pd.inv_depth[l]), _p(zt), _p(zr), _p(cam), _p(sic),
                               _p(r), _p(Jt))
        scale_r = max(1.0, np.abs(r[2:]).max())
        worst_r = max(worst_r, np.abs(r[:2] - r[2:]).max() / scale_r)
        worst_j = max(worst_j, np.abs(Jt[:12] - Jt[12:]).max() / np.abs(Jt[12:]).max())
        assert np.abs(Jt[:12]).max() > 0
    assert worst_r < 1e-11 and worst_j < 1e-12, (worst_r, worst_j)


def test_plus_and_log(hc):
    rng = np.random.RandomState(1)
    for _ in range(20):
        s = rng.randn(16)
        s[:4] /= np.linalg.norm(s[:4])
        d = rng.randn(15) * 0.05
        out = np.zeros(16)
        hc.hc_state_plus(_p(s), _p(d), _p(out))
        np.testing.assert_allclose(out, bo.state_plus(s, d), rtol=0, atol=1e-15)
        w = np.zeros(3)
        hc.hc_logmap(_p(s[:4].copy()), _p(w))
        np.testing.assert_allclose(w, bs.qlog(s[:4]), atol=1e-14)
