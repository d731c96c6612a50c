"""Independent numpy model of the reference's Initializer (row f3), written from
/root/reference/xrslam/src/xrslam/core/initializer.cpp:22-571 -- not from the C++ host pipeline it checks.

The C++ pipeline logs what every decision of an initialisation attempt looked at (XRSLAM_AMD_DUMP_INIT, csrc/host/ba_dump.hpp:
InitLogger); each function here re-derives the decision's answer from those inputs alone:

  keyframes(rec)                 mirror_keyframe_map (:22-76): which tracking-map frames become the init window, with which IMU samples
  two_view_models(rec)           init_sfm (:196-222): the model matrices explain the matches; the eight (R, T) hypotheses ARE decompositions
                                 of them (H ~ R + T n^T, E ~ [T]x R), in the reference's order and signs
  triangulation_vote(rec)        init_sfm [1.1] (:224-275): two-view DLT of every match under every hypothesis, cheirality / depth gate,
                                 counts, summed scores and the vote (kept quirk: a SUM of errors competes across hypotheses)
  gyro_bias(rec)                 solve_gyro_bias (:399-424)
  gravity_scale_velocity(rec)    solve_gravity_scale_velocity (:426-470), through numpy's least squares instead of a pivoted QR
  refine_via_gravity(rec)        refine_scale_velocity_via_gravity (:472-529), one damped step on the sphere |g| = 9.80665
  apply_init(rec)                apply_init (:545-571): gravity onto -z, metric scale, velocities
"""
import json

import numpy as np

GRAVITY_NOMINAL = 9.80665


def read(path):
    with open(path) as fh:
        return [json.loads(ln) for ln in fh if ln.strip()]


# ------------------------------------------------------------------------------------------------ quaternions (x, y, z, w)
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def qrot(q, v):
    return qmat(q) @ v


def qmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def logmap(q):
    """Eigen::AngleAxisd(q): angle * axis with the angle in [0, pi] (lie_algebra.h:20-23)"""
    q = np.array(q, float)
    if q[3] < 0:
        q = -q
    n = np.linalg.norm(q[:3])
    if n < 1e-300:
        return np.zeros(3)
    return 2.0 * np.arctan2(n, q[3]) * q[:3] / n


def from_two_vectors(a, b):
    """Quaterniond::FromTwoVectors for vectors that are not opposite"""
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    c = a @ b
    axis = np.cross(a, b)
    s = np.sqrt((1.0 + c) * 2.0)
    return np.r_[axis / s, 0.5 * s]


def tangential_basis(x):
    """s2_tangential_basis (geometry/lie_algebra.cpp:47-56): b1 = normalized(x cross e), e the axis where |x| is smallest... the log
    carries the answer of the solve, which depends on span(b1, b2) only -- any orthonormal basis of the tangent plane gives the same
    gravity update (the 2-vector dg changes with the basis, Tg dg does not)."""
    x = x / np.linalg.norm(x)
    e = np.eye(3)[np.argmin(np.abs(x))]
    b1 = np.cross(x, e)
    b1 /= np.linalg.norm(b1)
    b2 = np.cross(x, b1)
    return np.stack([b1, b2], 1)


# ------------------------------------------------------------------------------------------------ mirror_keyframe_map
def keyframes(rec):
    """-> (picked ids, IMU sample counts, first / last sample times of every picked frame's interval), or None when the tracking map is
    too short.  The init window is keyframe_num frames, keyframe_gap apart, ending at init_frame_id; a picked frame's interval is the
    concatenation of the samples of every tracking-map frame after the previous pick, up to itself (:55-72)."""
    ids = [int(v) for v in rec["ft_ids"]]
    last = ids.index(int(rec["init_frame_id"]))
    num, gap = int(rec["keyframe_num"]), int(rec["keyframe_gap"])
    dist = gap * (num - 1)
    if last < dist:
        return None
    idx = [last - dist + i * gap for i in range(num)]
    picked = [ids[i] for i in idx]
    samples, t0, t1 = [], [], []
    for j, i in enumerate(idx):
        if j == 0:   # the first picked frame keeps its own interval (a clone of the tracking map's frame)
            samples.append(int(rec["ft_samples"][i]))
            t0.append(rec["ft_t0"][i])
            t1.append(rec["ft_t1"][i])
            continue
        rng = range(idx[j - 1] + 1, i + 1)
        samples.append(int(sum(rec["ft_samples"][f] for f in rng)))
        with_data = [f for f in rng if rec["ft_samples"][f] > 0]
        t0.append(rec["ft_t0"][with_data[0]] if with_data else -1.0)
        t1.append(rec["ft_t1"][with_data[-1]] if with_data else -1.0)
    return picked, samples, t0, t1


# ------------------------------------------------------------------------------------------------ init_sfm
def two_view_models(rec):
    """What must hold between the logged matrices and the eight hypotheses (initializer.cpp:196-222): hypotheses 0..3 are
    (RH1, +-TH1), (RH2, +-TH2), 4..7 are (RE1, +-TE), (RE2, +-TE); every R a rotation, every T a unit vector; E ~ [T]x R for both
    twisted-pair members; H ~ RH + TH' nH^T with TH' parallel to TH (Malis-Vargas: H is normalised by its middle singular value).
    -> dict of residuals (all should be ~1e-9) and the inlier fractions of the two models at the RANSAC threshold 0.7 / fx."""
    Rs = np.array(rec["Rs"]).reshape(8, 3, 3)
    Ts = np.array(rec["Ts"]).reshape(8, 3)
    H = np.array(rec["H"]).reshape(3, 3)
    E = np.array(rec["E"]).reshape(3, 3)
    nH = np.array(rec["nH"]).reshape(2, 3)
    pi = np.array(rec["pi"]).reshape(-1, 2)
    pj = np.array(rec["pj"]).reshape(-1, 2)
    out = {}
    out["rotation_defect"] = max(max(np.abs(R @ R.T - np.eye(3)).max(), abs(np.linalg.det(R) - 1.0)) for R in Rs)
    out["unit_T_defect"] = max(abs(np.linalg.norm(t) - 1.0) for t in Ts)
    out["pairing_defect"] = max(np.abs(Rs[0] - Rs[1]).max(), np.abs(Rs[2] - Rs[3]).max(), np.abs(Rs[4] - Rs[5]).max(),
                                np.abs(Rs[6] - Rs[7]).max(), np.abs(Ts[0] + Ts[1]).max(), np.abs(Ts[2] + Ts[3]).max(),
                                np.abs(Ts[4] + Ts[5]).max(), np.abs(Ts[6] + Ts[7]).max(), np.abs(Ts[4] - Ts[6]).max())

    def hat(t):
        return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])

    def upto_scale(A, B):   # min over s of |A / |A| - s B / |B||, s = +-1
        A, B = A / np.linalg.norm(A), B / np.linalg.norm(B)
        return min(np.abs(A - B).max(), np.abs(A + B).max())
    out["essential_defect"] = max(upto_scale(E, hat(Ts[4]) @ Rs[4]), upto_scale(E, hat(Ts[6]) @ Rs[6]))
    # H / sigma_2 = R + t n^T with t parallel to the logged unit translation (sign of H: det > 0)
    Hn = H / np.linalg.svd(H, compute_uv=False)[1]
    if np.linalg.det(Hn) < 0:
        Hn = -Hn
    defects = []
    for R, T, n in ((Rs[0], Ts[0], nH[0]), (Rs[2], Ts[2], nH[1])):
        M = Hn - R                                  # = t n^T, rank one
        t = M @ n / (n @ n)
        defects.append(np.abs(M - np.outer(t, n)).max())
        defects.append(np.linalg.norm(np.cross(t, T)) / max(np.linalg.norm(t), 1e-300))   # t parallel to T
    out["homography_defect"] = max(defects)
    thr = 0.7 / rec["fx"]
    # inliers: symmetric transfer / epipolar distances (geometry/homography.h:17-21, essential.h:15-20) against chi2 thresholds
    a = np.c_[pi, np.ones(len(pi))]
    b = np.c_[pj, np.ones(len(pj))]
    hb = a @ H.T
    ha = b @ np.linalg.inv(H).T
    eh = ((hb[:, :2] / hb[:, 2:] - pj) ** 2).sum(1) + ((ha[:, :2] / ha[:, 2:] - pi) ** 2).sum(1)
    out["homography_inliers"] = float((eh < 5.99 * thr * thr * 2).mean())
    l2 = a @ E.T
    l1 = b @ E
    num = (b * l2).sum(1) ** 2
    ee = num / (l2[:, 0] ** 2 + l2[:, 1] ** 2) + num / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
    out["essential_inliers"] = float((ee < 3.84 * thr * thr * 2).mean())
    return out


def triangulate_two_view(P1, P2, z1, z2):
    """triangulate_point(P1, P2, z1, z2) (geometry/stereo.h:72-82): null vector of the 4x4 DLT system"""
    A = np.stack([z1[0] * P1[2] - z1[2] * P1[0], z1[1] * P1[2] - z1[2] * P1[1],
                  z2[0] * P2[2] - z2[2] * P2[0], z2[1] * P2[2] - z2[2] * P2[1]])
    return np.linalg.svd(A)[2][-1]


def triangulation_vote(rec):
    """-> (counts[8], scores[8], best, points of the best hypothesis [n, 3], status [n])   (initializer.cpp:224-275)"""
    Rs = np.array(rec["Rs"]).reshape(8, 3, 3)
    Ts = np.array(rec["Ts"]).reshape(8, 3)
    pi = np.array(rec["pi"]).reshape(-1, 2)
    pj = np.array(rec["pj"]).reshape(-1, 2)
    min_tri = int(rec["min_triangulation"])
    P1 = np.c_[np.eye(3), np.zeros(3)]
    counts, scores, pts, status = [], [], [], []
    best = 0
    for h in range(8):
        P2 = np.c_[Rs[h], Ts[h]]
        c, s = 0, 0.0
        p_h, st_h = np.zeros((len(pi), 3)), np.zeros(len(pi), int)
        for k in range(len(pi)):
            q = triangulate_two_view(P1, P2, np.r_[pi[k], 1.0], np.r_[pj[k], 1.0])
            q1, q2 = P1 @ q, P2 @ q
            if q1[2] * q[3] > 0 and q2[2] * q[3] > 0 and q1[2] / q[3] < 100 and q2[2] / q[3] < 100:
                p_h[k] = q[:3] / q[3]
                st_h[k] = 1
                c += 1
                s += 0.5 * (((q1[:2] / q1[2] - pi[k]) ** 2).sum() + ((q2[:2] / q2[2] - pj[k]) ** 2).sum())
        counts.append(c)
        scores.append(s)
        pts.append(p_h)
        status.append(st_h)
        if counts[h] > min_tri and scores[h] < scores[best]:
            best = h
        elif counts[h] > counts[best]:
            best = h
    return counts, scores, best, pts[best], status[best]


# ------------------------------------------------------------------------------------------------ init_imu
def gyro_bias(rec):
    """min over bg of sum |log((q_i dq)^-1 q_j) - dq_dbg bg|^2 through its 3x3 normal equations (:399-424)"""
    qi = np.array(rec["q_i"]).reshape(-1, 4)
    qj = np.array(rec["q_j"]).reshape(-1, 4)
    dq = np.array(rec["dq"]).reshape(-1, 4)
    J = np.array(rec["dq_dbg"]).reshape(-1, 3, 3)
    A, b = np.zeros((3, 3)), np.zeros(3)
    for k in range(len(qi)):
        A += J[k].T @ J[k]
        b += J[k].T @ logmap(qmul(qconj(qmul(qi[k], dq[k])), qj[k]))
    return np.linalg.lstsq(A, b, rcond=None)[0]


def _alignment_rows(rec, tangent_basis=None):
    dt = np.array(rec["dt"])
    dp = np.array(rec["dp"]).reshape(-1, 3)
    dv = np.array(rec["dv"]).reshape(-1, 3)
    ci = np.array(rec["cam_p_i"]).reshape(-1, 3)
    cj = np.array(rec["cam_p_j"]).reshape(-1, 3)
    qi = np.array(rec["body_q_i"]).reshape(-1, 4)
    qj = np.array(rec["body_q_j"]).reshape(-1, 4)
    pcs = np.array(rec["p_cs"])
    g0 = np.array(rec["gravity_before"])
    n = len(dt) + 1
    gc = 3 if tangent_basis is None else 2
    A = np.zeros((6 * (n - 1), gc + 1 + 3 * n))
    b = np.zeros(6 * (n - 1))
    G = np.eye(3) if tangent_basis is None else tangent_basis
    for i in range(n - 1):
        j = i + 1
        A[6 * i:6 * i + 3, :gc] = -0.5 * dt[i] ** 2 * G
        A[6 * i:6 * i + 3, gc] = cj[i] - ci[i]
        A[6 * i:6 * i + 3, gc + 1 + 3 * i:gc + 4 + 3 * i] = -dt[i] * np.eye(3)
        b[6 * i:6 * i + 3] = qrot(qi[i], dp[i]) + (qrot(qj[i], pcs) - qrot(qi[i], pcs))
        A[6 * i + 3:6 * i + 6, :gc] = -dt[i] * G
        A[6 * i + 3:6 * i + 6, gc + 1 + 3 * i:gc + 4 + 3 * i] = -np.eye(3)
        A[6 * i + 3:6 * i + 6, gc + 1 + 3 * j:gc + 4 + 3 * j] = np.eye(3)
        b[6 * i + 3:6 * i + 6] = qrot(qi[i], dv[i])
        if tangent_basis is not None:
            b[6 * i:6 * i + 3] += 0.5 * dt[i] ** 2 * g0
            b[6 * i + 3:6 * i + 6] += dt[i] * g0
    return A, b, n


def gravity_scale_velocity(rec):
    """-> (gravity, scale, velocities [n, 3])   (:426-470)"""
    A, b, n = _alignment_rows(rec)
    x = np.linalg.lstsq(A, b, rcond=None)[0]
    g = x[:3] / np.linalg.norm(x[:3]) * GRAVITY_NOMINAL
    return g, x[3], x[4:].reshape(n, 3)


def refine_via_gravity(rec, damp=0.1):
    """-> (gravity, scale, velocities)   (:472-529): one step, gravity moves in its tangent plane and is renormalised"""
    g0 = np.array(rec["gravity_before"])
    Tg = tangential_basis(g0)
    A, b, n = _alignment_rows(rec, Tg)
    x = np.linalg.lstsq(A, b, rcond=None)[0]
    g = g0 + damp * Tg @ x[:2]
    g = g / np.linalg.norm(g) * GRAVITY_NOMINAL
    return g, x[2], x[3:].reshape(n, 3)


def apply_init(rec):
    """-> (imu poses after [n, 7] as q xyzw + p, velocities after [n, 3])   (:545-571)"""
    g = np.array(rec["gravity"])
    q = from_two_vectors(g, np.array([0.0, 0.0, -GRAVITY_NOMINAL]))
    before = np.array(rec["imu_pose_before"]).reshape(-1, 7)
    vel = np.array(rec["velocities"]).reshape(-1, 3)
    poses, vs = [], []
    for k in range(len(before)):
        qq = qmul(q, before[k, :4])
        pp = rec["scale"] * qrot(q, before[k, 4:])
        poses.append(np.r_[qq, pp])
        vs.append(qrot(q, vel[k]))
    return np.array(poses), np.array(vs)
