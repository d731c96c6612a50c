"""Synthetic S1 "EuRoC MH_01-like" stream: a textured box room rendered through the EuRoC cam0 pinhole
model along the analytic trajectory, with 200 Hz IMU.  Seeded, no external data."""
import os

import numpy as np

from .trajectory import Trajectory, qmat, qmul, qrot

K_EUROC = (458.654, 457.296, 367.215, 248.375)
Q_BC = np.array([-7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01])
Q_BC = Q_BC / np.linalg.norm(Q_BC)
P_BC = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])


def _noise_texture(n, seed, octaves=7, mean=110.0, sigma=45.0):
    rng = np.random.RandomState(seed)
    acc = np.zeros((n, n))
    amp = 1.0
    for o in range(octaves):
        g = max(4, n >> (octaves - o))
        grid = rng.randn(g + 1, g + 1)
        xs = np.linspace(0, g - 1e-6, n)
        i0 = xs.astype(int)
        f = xs - i0
        rows = grid[i0] * (1 - f)[:, None] + grid[i0 + 1] * f[:, None]
        acc += amp * (rows[:, i0] * (1 - f)[None, :] + rows[:, i0 + 1] * f[None, :])
        amp *= 0.8
    acc = (acc - acc.mean()) / acc.std()
    return np.clip(mean + sigma * acc, 0, 255)


class BoxRoom:
    def __init__(self, half=(6.0, 6.0, 3.5), texels_per_m=110.0, seed=1):
        self.half = np.array(half)
        self.tpm = texels_per_m
        n = int(np.ceil(2 * max(half) * texels_per_m)) + 2
        self.tex = np.stack([_noise_texture(n, seed * 10 + k) for k in range(6)])   # face = 2*axis + (positive side)
        self.n = n

    def render(self, q_wc, p_wc, w=752, h=480, K=K_EUROC, dist=None, obj=None):
        """dist = (k1, k2, p1, p2): the image is what a radial-tangential lens would record (each pixel's ray is the
        undistorted direction whose distorted projection lands on that pixel); dist = ("equidistant", k1, k2, k3, k4):
        what a fisheye lens of the equidistant model would record."""
        fx, fy, cx, cy = K
        xs = (np.arange(w) - cx) / fx
        ys = (np.arange(h) - cy) / fy
        gx, gy = np.broadcast_arrays(xs[None, :], ys[:, None])
        if dist is not None and len(dist) == 5 and dist[0] == "equidistant":
            # ("equidistant", k1, k2, k3, k4): the fisheye model of the reference's TUM-VI reader
            # (xrslam-extra/include/xrslam/extra/image_undistorter.h:67-84): r_d = theta (1 + k1 theta^2 + ... + k4 theta^8)
            k1, k2, k3, k4 = dist[1:]
            rd = np.sqrt(gx * gx + gy * gy)
            th = rd.copy()
            for _ in range(20):                        # Newton on theta_d(theta) = r_d
                t2 = th * th
                f = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - rd
                df = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)))
                th = th - f / df
            sc = np.where(rd > 1e-12, np.tan(th) / np.where(rd > 1e-12, rd, 1.0), 1.0)
            gx, gy = gx * sc, gy * sc
        elif dist is not None:
            k1, k2, p1, p2 = dist
            xd, yd = gx, gy
            x, y = xd.copy(), yd.copy()
            for _ in range(12):                        # fixed-point inversion of the distortion model
                r2 = x * x + y * y
                kr = 1 + (k2 * r2 + k1) * r2
                dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
                dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
                x, y = (xd - dx) / kr, (yd - dy) / kr
            gx, gy = x, y
        d_cam = np.stack([gx, gy, np.ones((h, w))], axis=-1)
        d = d_cam @ qmat(q_wc).T                       # world ray directions
        with np.errstate(divide="ignore", invalid="ignore"):
            side = np.where(d >= 0, 1.0, -1.0)
            t_axis = (side * self.half - p_wc) / d
        t_axis = np.where(np.isfinite(t_axis) & (t_axis > 0), t_axis, np.inf)
        axis = np.argmin(t_axis, axis=-1)
        t = np.take_along_axis(t_axis, axis[..., None], axis=-1)[..., 0]
        hit = p_wc + d * t[..., None]
        face = 2 * axis + (np.take_along_axis(side, axis[..., None], axis=-1)[..., 0] > 0)
        ua = np.where(axis == 0, 1, 0)                 # in-plane axes
        va = np.where(axis == 2, 1, 2)
        u = (np.take_along_axis(hit, ua[..., None], axis=-1)[..., 0] + self.half.max()) * self.tpm
        v = (np.take_along_axis(hit, va[..., None], axis=-1)[..., 0] + self.half.max()) * self.tpm
        u = np.clip(u, 0, self.n - 1.001)
        v = np.clip(v, 0, self.n - 1.001)
        u0 = u.astype(int)
        v0 = v.astype(int)
        fu = u - u0
        fv = v - v0
        T = self.tex
        val = (T[face, v0, u0] * (1 - fu) * (1 - fv) + T[face, v0, u0 + 1] * fu * (1 - fv) + T[face, v0 + 1, u0] * (1 - fu) * fv +
               T[face, v0 + 1, u0 + 1] * fu * fv)
        if obj is not None:
            val = self._paint_object(val, p_wc, d, t, obj)
        return np.clip(np.rint(val), 0, 255).astype(np.uint8)

    def _paint_object(self, val, p_wc, d, t_room, obj):
        """A textured cuboid (centre, half extents) in front of the walls: the moving object of the RD-VIO tests."""
        c, hs = np.asarray(obj[0], float), np.asarray(obj[1], float)
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (c - hs - p_wc) / d
            t2 = (c + hs - p_wc) / d
        tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
        t_in, t_out = np.nanmax(tn, axis=-1), np.nanmin(tf, axis=-1)
        hit_box = (t_in < t_out) & (t_in > 0) & (t_in < t_room)
        if not hit_box.any():
            return val
        axis = np.argmax(tn, axis=-1)
        hit = p_wc + d * t_in[..., None] - c                       # in the object's frame: texture sticks to it
        ua = np.where(axis == 0, 1, 0)
        va = np.where(axis == 2, 1, 2)
        u = (np.take_along_axis(hit, ua[..., None], axis=-1)[..., 0] + 2.0) * self.tpm
        v = (np.take_along_axis(hit, va[..., None], axis=-1)[..., 0] + 2.0) * self.tpm
        u = np.clip(u, 0, self.n - 1.001)
        v = np.clip(v, 0, self.n - 1.001)
        u0, v0 = u.astype(int), v.astype(int)
        fu, fv = u - u0, v - v0
        To = np.roll(self.tex, 977, axis=2)
        face = (axis + 3) % 6
        vo = (To[face, v0, u0] * (1 - fu) * (1 - fv) + To[face, v0, u0 + 1] * fu * (1 - fv) + To[face, v0 + 1, u0] * (1 - fu) * fv +
              To[face, v0 + 1, u0 + 1] * fu * fv)
        return np.where(hit_box, vo, val)


_RENDER_JOB = None   # (room, traj, cam_t, w, h, K, dist, moving_object): inherited by the forked render workers


def _render_range(lo_hi):
    room, traj, cam_t, w, h, K, dist, moving_object = _RENDER_JOB
    lo, hi = lo_hi
    out = np.zeros((hi - lo, h, w), np.uint8)
    for i in range(lo, hi):
        t = cam_t[i]
        q, p = traj.q(t), traj.p(t)
        obj = moving_object(t) if moving_object is not None else None
        out[i - lo] = room.render(qmul(q, Q_BC), p + qrot(q, P_BC), w, h, K, dist, obj)
    return lo, out


def make_sequence(n_frames=120, w=752, h=480, seed=1, cam_hz=20.0, imu_hz=200.0, t0=5.0, noise=True, traj=None, K=K_EUROC, dist=None, moving_object=None,
                  workers=None):
    """Returns dict(frames uint8 [n,h,w], cam_t [n], imu [m,7] (t, w, a), states [n,16] ground truth body states).
    A sequence of n frames is a prefix of every longer one with the same arguments (the random draws are sequential in
    time and the frames are rendered independently).  workers > 1: the frames are rendered by that many forked processes
    (same pixels; call it before anything in the process has touched the GPU runtime)."""
    if workers is None:   # XRSLAM_AMD_RENDER_WORKERS: the default for callers that do not say (tests/conftest.py sets it on GPU-less hosts)
        workers = int(os.environ.get("XRSLAM_AMD_RENDER_WORKERS", "1"))
    rng = np.random.RandomState(seed)
    traj = traj or Trajectory()
    room = BoxRoom(seed=seed)
    bg = rng.randn(3) * 1e-3
    ba = rng.randn(3) * 1e-2
    ratio = int(round(imu_hz / cam_hz))
    n_imu = (n_frames - 1) * ratio + ratio + 1
    imu = np.zeros((n_imu, 7))
    sg, sa = (1.7e-4 * np.sqrt(imu_hz), 2.0e-3 * np.sqrt(imu_hz)) if noise else (0.0, 0.0)
    for k in range(n_imu):
        t = t0 + k / imu_hz
        wv, av = traj.imu(t, bg, ba)
        imu[k] = np.concatenate([[t], wv + rng.randn(3) * sg, av + rng.randn(3) * sa])
    frames = np.zeros((n_frames, h, w), np.uint8)
    cam_t = t0 + np.arange(n_frames) / cam_hz
    states = np.zeros((n_frames, 16))
    for i, t in enumerate(cam_t):
        states[i] = np.concatenate([traj.q(t), traj.p(t), traj.v(t), bg, ba])
    global _RENDER_JOB
    _RENDER_JOB = (room, traj, cam_t, w, h, K, dist, moving_object)   # moving_object: callable t -> (centre, half extents) or None
    try:
        workers = max(1, min(int(workers), n_frames // 8))
        if workers > 1:
            import multiprocessing as mp
            chunk = max(1, (n_frames + 4 * workers - 1) // (4 * workers))
            jobs = [(lo, min(n_frames, lo + chunk)) for lo in range(0, n_frames, chunk)]
            with mp.get_context("fork").Pool(workers) as pool:
                for lo, part in pool.imap_unordered(_render_range, jobs):
                    frames[lo:lo + len(part)] = part
        else:
            frames[:] = _render_range((0, n_frames))[1]
    finally:
        _RENDER_JOB = None
    return dict(frames=frames, cam_t=cam_t, imu=imu, states=states, bg=bg, ba=ba)
