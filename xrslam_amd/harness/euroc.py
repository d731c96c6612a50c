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
