"""Quaternion helpers (x, y, z, w) and the analytic body trajectory used by every synthetic workload."""
import numpy as np

GRAVITY = 9.80665


def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def qrot(q, v):
    u = q[:3]
    uv = 2 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


def qexp(w):
    a = np.linalg.norm(w)
    if a < 1e-12:
        return np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0]) / np.sqrt(1 + 0.25 * a * a)
    s = np.sin(0.5 * a) / a
    return np.array([s * w[0], s * w[1], s * w[2], np.cos(0.5 * a)])


def qlog(q):
    n = np.linalg.norm(q[:3])
    if n < 1e-300:
        return np.zeros(3)
    ang = 2 * np.arctan2(n, abs(q[3]))
    return q[:3] / (n if q[3] >= 0 else -n) * ang


def qmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Trajectory:
    """Figure-eight position, gentle attitude oscillation (stream S1)."""

    def __init__(self, amp=1.5, speed=0.6, rot=0.35):
        self.amp, self.speed, self.rot = amp, speed, rot

    def p(self, t):
        s = self.speed
        return np.array([self.amp * np.sin(s * t), self.amp * np.sin(2 * s * t) * 0.5, 0.3 * np.sin(0.7 * s * t)])

    def v(self, t, h=1e-5):
        return (self.p(t + h) - self.p(t - h)) / (2 * h)

    def a(self, t, h=1e-4):
        return (self.p(t + h) - 2 * self.p(t) + self.p(t - h)) / (h * h)

    def q(self, t):
        yaw = self.rot * np.sin(0.8 * self.speed * t)
        pitch = 0.5 * self.rot * np.sin(1.3 * self.speed * t + 0.3)
        roll = 0.4 * self.rot * np.sin(0.9 * self.speed * t + 1.1)
        return qmul(qmul(qexp(np.array([0, 0, yaw])), qexp(np.array([0, pitch, 0]))), qexp(np.array([roll, 0, 0])))

    def w_body(self, t, h=1e-5):
        return qlog(qmul(qconj(self.q(t - h)), self.q(t + h))) / (2 * h)

    def imu(self, t, bg, ba):
        R = qmat(self.q(t))
        acc = R.T @ (self.a(t) - np.array([0, 0, -GRAVITY]))
        return self.w_body(t) + bg, acc + ba
