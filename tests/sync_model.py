"""Independent Python model of the reference's sensor synchronisation, XRSLAM::Detail (core/detail.cpp:15-28, 46-177):
gyroscope / accelerometer samples arrive on their own clocks, every accelerometer sample becomes one IMU datum with the angular
rate interpolated to its time stamp, IMU data are attached to the oldest waiting frame until one is later than that frame (which
releases the frame to the feature tracker), and the pose answered for a time stamp is the tracker's latest state propagated over the
IMU data received since.  Written from the reference source, not from xrslam_amd/csrc/host/pipeline.hpp; tests/test_sync_model.py
feeds it the same events as the C++ pipeline and compares."""
import collections

import numpy as np

GRAVITY = np.array([0.0, 0.0, -9.80665])      # XRSLAM_GRAVITY_NOMINAL


def q_mul(a, b):      # xyzw
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def q_rot(q, v):
    qv = np.array([v[0], v[1], v[2], 0.0])
    qc = np.array([-q[0], -q[1], -q[2], q[3]])
    return q_mul(q_mul(q, qv), qc)[:3]


def expmap(w):        # geometry/lie_algebra.h:13-18: AngleAxis(|w|, w / |w|) as a quaternion
    th = np.linalg.norm(w)
    if th == 0.0:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = w / th
    return np.array([*(np.sin(0.5 * th) * ax), np.cos(0.5 * th)])


class DetailModel:
    def __init__(self, q_bo=(0, 0, 0, 1), p_bo=(0, 0, 0)):
        self.gyros = collections.deque()      # (t, w)
        self.accs = collections.deque()       # (t, a)
        self.imus = collections.deque()       # (t, w, a) not yet attached to a frame
        self.frontal = collections.deque()    # (t, w, a) since the tracker's latest state
        self.frames = collections.deque()     # [t, [samples]]
        self.released = []                    # (t, samples) in the order the feature tracker receives them
        self.q_bo, self.p_bo = np.array(q_bo, float), np.array(p_bo, float)

    # detail.cpp:46-70
    def track_gyroscope(self, t, w):
        w = np.array(w, float)
        if self.accs:
            if t < self.accs[0][0]:
                self.gyros.clear()
            else:
                while self.accs and t >= self.accs[0][0]:
                    at, a = self.accs[0]
                    g0t, g0w = self.gyros[0]
                    lam = (at - g0t) / (t - g0t)
                    self.track_imu(at, g0w + lam * (w - g0w), a)
                    self.accs.popleft()
                if self.accs:
                    while self.gyros and self.gyros[0][0] < t:
                        self.gyros.popleft()
        self.gyros.append((t, w))

    # detail.cpp:72-101
    def track_accelerometer(self, t, a):
        a = np.array(a, float)
        if self.gyros and t >= self.gyros[0][0]:
            if t > self.gyros[-1][0]:
                while len(self.gyros) > 1:
                    self.gyros.popleft()
                self.accs.append((t, a))
            elif t == self.gyros[-1][0]:
                while len(self.gyros) > 1:
                    self.gyros.popleft()
                self.track_imu(t, self.gyros[0][1], a)
            else:
                while t >= self.gyros[1][0]:
                    self.gyros.popleft()
                (t0, w0), (t1, w1) = self.gyros[0], self.gyros[1]
                lam = (t - t0) / (t1 - t0)
                self.track_imu(t, w0 + lam * (w1 - w0), a)

    # detail.cpp:103-128 (the part that concerns synchronisation)
    def track_camera(self, t):
        self.frames.append([t, []])

    # detail.cpp:130-142
    def track_imu(self, t, w, a):
        self.frontal.append((t, w, a))
        self.imus.append((t, w, a))
        while self.imus and self.frames:
            if self.imus[0][0] <= self.frames[0][0]:
                self.frames[0][1].append(self.imus.popleft())
            else:
                f = self.frames.popleft()
                self.released.append((f[0], f[1]))

    # detail.cpp:15-28, 144-168, from the tracker's latest state (time, q xyzw, p, v, bg, ba)
    def predict_pose(self, t, state):
        st, q, p, v, bg, ba = state[0], *[np.array(x, float) for x in state[1:]]
        while self.frontal and self.frontal[0][0] <= st:
            self.frontal.popleft()
        for it, w, a in self.frontal:
            if it <= t:
                dt = it - st
                acc = GRAVITY + q_rot(q, a - ba)
                p = p + dt * v + 0.5 * dt * dt * acc
                v = v + dt * acc
                q = q_mul(q, expmap((w - bg) * dt))
                q = q / np.linalg.norm(q)
                st = it
        return q_mul(q, self.q_bo), p + q_rot(q, self.p_bo)
