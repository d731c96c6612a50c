"""Independent numpy model of the RD-VIO PARSAC machinery (row f4), written from
/root/reference/xrslam/src/xrslam/utility/parsac.h (:50-381), utility/imu_parsac.h (:9-415) and the verdict of
core/sliding_window_tracker.cpp:577-739 -- not from csrc/host/parsac.hpp, whose decisions it checks.

The C++ pipeline logs every PARSAC run (XRSLAM_AMD_DUMP_INIT: "parsac_run") with what it looked at -- the points that are bucketed,
track lengths, the prior bin confidences, the IMU-prior inlier mask -- and, per scored hypothesis, its inlier mask (the one thing
that needs the minimal solvers, which have their own tests).  From those alone `replay()` re-derives: the 20 x 20 bucketing and the
order of the occupied bins, the thresholded / normalised / accumulated prior, every hypothesis' score (float arithmetic as the
reference's: float accumulators, double products), which hypothesis takes over as the best, the adaptive iteration cap after each,
how many iterations run, the final mask / inlier count / give-up outcome, and the bin confidences written back for the next call.
`judge()` re-derives the verdict of judge_track_status from the logged epipolar distances."""
import numpy as np

F = np.float32


def bucket(pts, norm_scale=1.0, nx=20, ny=20):
    """BucketData (parsac.h:260-285): -> (data_to_valid, valid_to_bin, valid_size, bin locations [400, 2]); bins in first-touch order"""
    bin_h, bin_w = F(2 * norm_scale / ny), F(2 * norm_scale / nx)
    loc = []
    y = F(bin_h * F(0.5))
    for _ in range(ny):
        x = F(bin_w * F(0.5))
        for _ in range(nx):
            loc.append((float(x) - norm_scale, float(y) - norm_scale))
            x = F(x + bin_w)
        y = F(y + bin_h)
    bin_to_valid, valid_to_bin, valid_size, data_to_valid = {}, [], [], []
    for p in pts:
        bx = min(nx - 1, int(max(0.0, (p[0] + norm_scale) / float(bin_w))))     # (out-of-grid points are clamped by the pipeline)
        by = min(ny - 1, int(max(0.0, (p[1] + norm_scale) / float(bin_h))))
        b = bx + nx * by
        if b not in bin_to_valid:
            bin_to_valid[b] = len(valid_to_bin)
            valid_to_bin.append(b)
            valid_size.append(0)
        v = bin_to_valid[b]
        data_to_valid.append(v)
        valid_size[v] += 1
    return np.array(data_to_valid), valid_to_bin, valid_size, np.array(loc)


def accumulated_prior(bins_before, valid_to_bin, floor=0.5):
    """ThresholdAndNormalizeConfidences + AccumulateConfidences (parsac.h:325-352), float"""
    c = [max(F(floor), F(bins_before[b])) for b in valid_to_bin]
    s = F(0)
    for v in c:
        s = F(s + v)
    norm = F(1.0 / float(s))
    c = [F(v * norm) for v in c]
    acc = [F(0)]
    for v in c:
        acc.append(F(acc[-1] + v))
    n2 = F(F(1) / acc[-1])
    return [F(a * n2) for a in acc[:-1]] + [acc[-1]]


def score(inliers_per_bin, valid_size, valid_len, valid_to_bin, loc, dynamic_probability):
    """ComputeScore (parsac.h:198-243 / imu_parsac.h:240-290): -> (score, per-bin confidences); dynamic_probability None = plain variant"""
    nv = len(valid_size)
    conf = []
    csum, csq = F(0), F(0)
    sx = sy = 0.0
    for v in range(nv):
        c = F(F(inliers_per_bin[v]) / F(valid_size[v]))
        if dynamic_probability is not None:
            t = F(1 - dynamic_probability ** (0.10 * float(valid_len[v])))
            c = F(F(t * F(inliers_per_bin[v])) / F(valid_size[v]))
        conf.append(c)
        x = loc[valid_to_bin[v]]
        sx += x[0] * float(c)
        sy += x[1] * float(c)
        csum = F(csum + c)
        csq = F(csq + F(c * c))
    norm = F(F(1) / csum) if csum != 0 else F(np.inf)
    mx, my = sx * float(norm), sy * float(norm)
    cxx = cxy = cyy = F(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        for v in range(nv):
            x = loc[valid_to_bin[v]]
            dx, dy = x[0] - mx, x[1] - my
            cxx = F(float(cxx) + dx * dx * float(conf[v]))
            cxy = F(float(cxy) + dx * dy * float(conf[v]))
            cyy = F(float(cyy) + dy * dy * float(conf[v]))
        norm = F(csum / F(F(csum * csum) - csq))
        ratio = F(norm * np.sqrt(F(F(cxx * cyy) - F(cxy * cxy))))
        return F(ratio * csum), conf


def replay(rec):
    """-> dict of everything the run decided, derived from the logged inputs and the logged per-hypothesis inlier masks"""
    size, dof, imu = int(rec["size"]), int(rec["dof"]), bool(rec["imu"])
    pts = np.array(rec["grid_pts"]).reshape(-1, 2)
    d2v, v2b, vsize, loc = bucket(pts, rec["norm_scale"])
    nv = len(vsize)
    lens = np.array(rec["lens"]) if imu else None
    vlen = [F(0)] * nv
    if imu:
        acc_len = [F(0)] * nv
        for i, v in enumerate(d2v):
            acc_len[v] = F(acc_len[v] + F(lens[i]))
        vlen = [F(acc_len[v] / F(vsize[v])) for v in range(nv)]
    prior = np.array(rec["prior_mask"], int) if imu else None
    dyn = rec["dynamic_probability"] if imu else None
    K = np.log(max(1 - rec["confidence"], 1.0e-5))
    masks = np.array(rec["cand_masks"], int).reshape(-1, size) if len(rec["cand_masks"]) else np.zeros((0, size), int)
    iters = [int(v) for v in rec["cand_iter"]]
    iter_max = int(rec["max_iteration"])
    score_max = F(-np.finfo(np.float32).max) if imu else F(0)
    inlier_count, best, best_bins = 0, None, None
    scores, takes, caps = [], [], []
    last_take_iter = -1
    for k, m in enumerate(masks):
        assert iters[k] < iter_max, "a hypothesis was scored in an iteration the cap had already excluded"
        counted = int(m.sum()) if not imu else int((m & prior).sum())
        per_bin = np.bincount(d2v[m == 1], minlength=nv)
        sc, _ = score(per_bin, vsize, vlen, v2b, loc, dyn)
        take = bool(sc > score_max or (sc == score_max and counted > inlier_count))
        if take:
            score_max, inlier_count, best, best_bins = sc, counted, k, per_bin
            last_take_iter = iters[k]
            ratio = inlier_count / float(size)
            with np.errstate(divide="ignore"):
                n = K / np.log(1 - ratio ** 5) if ratio < 1 else 0.0
            if n < iter_max:
                iter_max = int(np.ceil(n))
        scores.append(float(sc))
        takes.append(take)
        caps.append(iter_max)
    gave_up = imu and inlier_count < dof
    bins_after = None
    if not gave_up:
        if best_bins is None:
            best_bins = np.zeros(nv, int)
        _, conf = score(best_bins, vsize, vlen, v2b, loc, dyn)
        bins_after = np.zeros(400, np.float32)
        for v, b in enumerate(v2b):
            bins_after[b] = conf[v]
    return {"nvalid": nv, "accumulated_prior": np.array(accumulated_prior(rec["bins_before"], v2b), np.float32),
            "scores": scores, "takes_over": takes, "iter_max_after": caps, "iterations_run": max(iter_max, last_take_iter + 1),   # the cap is tested at the top of an iteration: the one that lowers it still completes
            "gave_up": bool(gave_up), "inlier_count": inlier_count, "final_mask": None if best is None else masks[best],
            "bins_after": bins_after}


def judge(rec):
    """judge_track_status's verdict (sliding_window_tracker.cpp:700-739): the medians of the epipolar distances of the PnP consensus'
    inliers and outliers must differ by a factor 2; the 2D-2D threshold is their mean.  -> (separated, threshold)"""
    d_in, d_out = sorted(rec["d_in"]), sorted(rec["d_out"])
    if len(d_in) < 20 or len(d_out) < 20:
        return False, 0.0
    th1, th2 = d_in[int(len(d_in) * 0.5)], d_out[int(len(d_out) * 0.5)]
    if th2 < th1 * 2:
        return False, 0.0
    return True, (th1 + th2) / 2
