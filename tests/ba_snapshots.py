"""Frozen BA problems (SURVEY.md 8d "S4"): reader of the files xrslam_amd/csrc/host/ba_dump.hpp writes, and of the
committed fixtures under tests/golden/ba_snapshots/*.npz (made from them by tests/golden/make_ba_snapshots.py, together
with the oracle's full result: iterations, termination, costs, final states).  Test / bench infrastructure."""
import glob
import os

import numpy as np

from xrslam_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAP_DIR = os.path.join(ROOT, "tests", "golden", "ba_snapshots")
FIELDS = ("frame_state", "frame_fix", "cam_ext", "imu_ext", "sqrt_inv_cov", "inv_depth", "landmark_fix", "obs_tgt", "obs_ref",
          "obs_lm", "obs_z_tgt", "obs_z_ref", "rot_tgt", "rot_ref", "rot_z_tgt", "rot_z_ref", "imu_i", "imu_j", "imu_data",
          "prior_frames", "prior_sqrt_info", "prior_infovec", "prior_lin", "max_iterations")


def read_xrba(path):
    """-> dict of numpy arrays (the fields of xrhip_ba_problem)."""
    raw = open(path, "rb").read()
    assert raw[:5] == b"XRBA1", "not a BA snapshot: " + path
    F, L, M, MR, NI, NP, max_it, _ = np.frombuffer(raw, np.int32, 8, 8)
    off = [8 + 32]

    def take(dtype, n):
        a = np.frombuffer(raw, dtype, n, off[0]).copy()
        off[0] += a.nbytes
        return a
    d = {}
    d["frame_state"] = take(np.float64, 16 * F).reshape(F, 16)
    d["frame_fix"] = take(np.uint8, F)
    cq, cp, iq, ip = take(np.float64, 4), take(np.float64, 3), take(np.float64, 4), take(np.float64, 3)
    d["cam_ext"], d["imu_ext"] = np.concatenate([cq, cp]), np.concatenate([iq, ip])
    d["sqrt_inv_cov"] = take(np.float64, 2)
    d["inv_depth"] = take(np.float64, L)
    d["landmark_fix"] = take(np.uint8, L)
    d["obs_tgt"], d["obs_ref"], d["obs_lm"] = take(np.int32, M), take(np.int32, M), take(np.int32, M)
    d["obs_z_tgt"], d["obs_z_ref"] = take(np.float64, 3 * M).reshape(M, 3), take(np.float64, 3 * M).reshape(M, 3)
    d["rot_tgt"], d["rot_ref"] = take(np.int32, MR), take(np.int32, MR)
    d["rot_z_tgt"], d["rot_z_ref"] = take(np.float64, 3 * MR).reshape(MR, 3), take(np.float64, 3 * MR).reshape(MR, 3)
    d["imu_i"], d["imu_j"] = take(np.int32, NI), take(np.int32, NI)
    d["imu_data"] = take(np.float64, abi.IMU_DIM * NI).reshape(NI, abi.IMU_DIM)
    d["prior_frames"] = take(np.int32, NP)
    d["prior_sqrt_info"] = take(np.float64, 225 * NP * NP).reshape(15 * NP, 15 * NP)
    d["prior_infovec"] = take(np.float64, 15 * NP)
    d["prior_lin"] = take(np.float64, 16 * NP).reshape(NP, 16)
    d["max_iterations"] = np.int32(max_it)
    assert off[0] == len(raw), "trailing bytes in " + path
    return d


def to_problem(d):
    """dict of arrays -> abi.BaProblemData (a fresh copy: solves work in place)."""
    return abi.BaProblemData(
        d["frame_state"], d["frame_fix"], d["cam_ext"], d["imu_ext"], d["sqrt_inv_cov"], d["inv_depth"], d["landmark_fix"],
        obs=dict(tgt=d["obs_tgt"], ref=d["obs_ref"], lm=d["obs_lm"], z_tgt=d["obs_z_tgt"], z_ref=d["obs_z_ref"]),
        rot=dict(tgt=d["rot_tgt"], ref=d["rot_ref"], z_tgt=d["rot_z_tgt"], z_ref=d["rot_z_ref"]),
        imu=dict(i=d["imu_i"], j=d["imu_j"], data=d["imu_data"]),
        prior=dict(frames=d["prior_frames"], sqrt_info=d["prior_sqrt_info"], infovec=d["prior_infovec"], lin=d["prior_lin"]),
        max_iterations=int(d["max_iterations"]))


def load_all():
    """-> [(name, BaProblemData, expected dict)] for every committed fixture."""
    out = []
    for path in sorted(glob.glob(os.path.join(SNAP_DIR, "*.npz"))):
        z = np.load(path)
        d = {k: z[k] for k in FIELDS}
        exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
        out.append((os.path.splitext(os.path.basename(path))[0], to_problem(d), exp))
    return out
