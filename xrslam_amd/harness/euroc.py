"""Writes a synthetic sequence (harness.scene.make_sequence) as an ASL / EuRoC directory, the format the reference's
player reads (xrslam-pc/player/src/IO/euroc_dataset_reader.h:28-117) -- test data for xrslam-player."""
import os

import numpy as np


def write_euroc(seq, root, n_frames=None):
    """root = <dir>/mav0.  Ground truth (body pose, velocity, biases) is written at the camera time stamps."""
    from PIL import Image
    n = len(seq["cam_t"]) if n_frames is None else min(n_frames, len(seq["cam_t"]))
    os.makedirs(os.path.join(root, "cam0", "data"), exist_ok=True)
    os.makedirs(os.path.join(root, "imu0"), exist_ok=True)
    os.makedirs(os.path.join(root, "state_groundtruth_estimate0"), exist_ok=True)
    ns = lambda t: int(round(float(t) * 1e9))
    with open(os.path.join(root, "cam0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp [ns],filename\r\n")
        for i in range(n):
            name = "%019d.png" % ns(seq["cam_t"][i])
            Image.fromarray(seq["frames"][i]).save(os.path.join(root, "cam0", "data", name))
            f.write("%d,%s\r\n" % (ns(seq["cam_t"][i]), name))
    t_last = float(seq["cam_t"][n - 1])
    with open(os.path.join(root, "imu0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],"
                "a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\r\n")
        for r in seq["imu"]:
            if r[0] > t_last + 0.02:
                break
            f.write("%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\r\n" % ((ns(r[0]),) + tuple(r[1:7])))
    with open(os.path.join(root, "state_groundtruth_estimate0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], "
                "v_RS_R_x [m s^-1], v_RS_R_y [m s^-1], v_RS_R_z [m s^-1], b_w_RS_S_x [rad s^-1], b_w_RS_S_y [rad s^-1], "
                "b_w_RS_S_z [rad s^-1], b_a_RS_S_x [m s^-2], b_a_RS_S_y [m s^-2], b_a_RS_S_z [m s^-2]\r\n")
        for i in range(n):
            s = np.asarray(seq["states"][i], float)        # q xyzw, p, v, bg, ba
            q, p, v, bg, ba = s[0:4], s[4:7], s[7:10], s[10:13], s[13:16]
            vals = (ns(seq["cam_t"][i]),) + tuple(p) + (q[3], q[0], q[1], q[2]) + tuple(v) + tuple(bg) + tuple(ba)
            f.write(("%d" + ",%.17g" * 16 + "\r\n") % vals)
    return root


def read_euroc(root, max_frames=None):
    """root = <dir>/mav0 of an ASL / EuRoC sequence -> the dict harness.scene.make_sequence returns (frames uint8 [n,h,w]
    AS RECORDED -- rectify with XRSLAMAmdSetDeviceUndistort --, cam_t [s], imu [m,7] = t, gyroscope, accelerometer,
    states [n,16] ground truth interpolated to the camera times where state_groundtruth_estimate0 exists, else zeros).
    The layout the reference's reader parses (xrslam-pc/player/src/IO/euroc_dataset_reader.h:28-117)."""
    from PIL import Image
    rows = [ln.strip().split(",") for ln in open(os.path.join(root, "cam0", "data.csv")) if ln.strip() and not ln.startswith("#")]
    if max_frames is not None:
        rows = rows[:max_frames]
    cam_t = np.array([int(r[0]) * 1e-9 for r in rows])
    first = np.asarray(Image.open(os.path.join(root, "cam0", "data", rows[0][1].strip())))
    frames = np.empty((len(rows),) + first.shape[:2], np.uint8)
    for i, r in enumerate(rows):
        im = np.asarray(Image.open(os.path.join(root, "cam0", "data", r[1].strip())))
        if im.dtype == np.uint16:
            im = (im >> 8).astype(np.uint8)                      # cv::imread(IMREAD_GRAYSCALE) strips 16-bit samples to their high byte
        frames[i] = im if im.ndim == 2 else im[..., 0]
    imu = np.array([[float(v) for v in ln.strip().split(",")[:7]] for ln in open(os.path.join(root, "imu0", "data.csv"))
                    if ln.strip() and not ln.startswith("#")])
    imu[:, 0] *= 1e-9
    imu = imu[imu[:, 0] <= cam_t[-1] + 0.05]
    states = np.zeros((len(rows), 16))
    gt_path = os.path.join(root, "state_groundtruth_estimate0", "data.csv")
    if os.path.exists(gt_path):
        g = np.array([[float(v) for v in ln.strip().split(",")[:17]] for ln in open(gt_path) if ln.strip() and not ln.startswith("#")])
        tg = g[:, 0] * 1e-9
        idx = np.clip(np.searchsorted(tg, cam_t), 1, len(tg) - 1)
        idx = np.where(np.abs(tg[idx - 1] - cam_t) < np.abs(tg[idx] - cam_t), idx - 1, idx)   # nearest sample (200 Hz)
        r = g[idx]
        states[:, 0:3], states[:, 3] = r[:, 5:8], r[:, 4]         # ASL stores the quaternion w first
        states[:, 4:7], states[:, 7:10], states[:, 10:13], states[:, 13:16] = r[:, 1:4], r[:, 8:11], r[:, 11:14], r[:, 14:17]
        states[np.abs(tg[idx] - cam_t) > 0.01] = 0.0              # no ground truth near this frame
    return dict(frames=frames, cam_t=cam_t, imu=imu, states=states)
