"""CPU tests of the RD-VIO outlier filters (xrslam_amd/csrc/host/epnp.hpp, parsac.hpp) on synthetic
static + moving point sets.  The reference gets EPnP from OpenCV and was never run here ("parity unpinned", see the
file headers): these checks pin the behaviour that matters -- the static scene is kept, the moving object is dropped."""
import ctypes as C

import numpy as np
import pytest

from tests import ba_synth as bs
from tests.test_host_geometry import gh  # noqa: F401  (fixture: builds tests/host_check/geom_host.cpp)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pose(rng, angle=0.3, trans=0.5):
    ax = rng.randn(3)
    ax /= np.linalg.norm(ax)
    return bs.qmat(bs.qexp(ax * angle)), rng.randn(3) * trans


def _scene(rng, n, R, t):
    """points in front of the camera described by x_cam = R X + t"""
    Xc = np.c_[rng.uniform(-0.7, 0.7, n), rng.uniform(-0.5, 0.5, n), np.ones(n)] * rng.uniform(2.0, 8.0, n)[:, None]
    X = (Xc - t) @ R               # world points: R^T (Xc - t)
    x = Xc[:, :2] / Xc[:, 2:]
    return np.ascontiguousarray(X), np.ascontiguousarray(x)


def test_epnp_recovers_the_pose(gh):
    rng = np.random.RandomState(1)
    for n in (6, 6, 6, 12, 40):
        for _ in range(10):
            R, t = _pose(rng)
            X, x = _scene(rng, n, R, t)
            Ro, to = np.zeros(9), np.zeros(3)
            gh.gh_epnp(_p(X), _p(x), n, _p(Ro), _p(to))
            Ro = Ro.reshape(3, 3)
            # float32 inputs and outputs, like the reference's wrapper around OpenCV
            assert np.abs(Ro - R).max() < 2e-4 and np.abs(to - t).max() < 2e-3, (n, np.abs(Ro - R).max(), np.abs(to - t).max())
            assert abs(np.linalg.det(Ro) - 1) < 1e-5


def test_imu_parsac_drops_the_moving_object(gh):
    rng = np.random.RandomState(2)
    for trial in range(5):
        gh.gh_parsac_reset()
        R, t = _pose(rng, 0.2, 0.4)
        n_static, n_moving = 120, 30
        X, x = _scene(rng, n_static + n_moving, R, t)
        # the object occupies one image region and has moved since its landmarks were triangulated
        x[n_static:] = np.c_[rng.uniform(0.2, 0.5, n_moving), rng.uniform(-0.3, 0.0, n_moving)]
        depth = rng.uniform(2.0, 3.0, n_moving)
        Xc = np.c_[x[n_static:] * depth[:, None], depth] + np.array([0.15, -0.1, 0.0])      # displaced by ~0.05 normalised units
        X[n_static:] = (Xc - t) @ R
        x += rng.randn(*x.shape) * 0.3 / 458.0
        lens = np.full(len(X), 8, np.int64)
        # IMU prediction: the true pose perturbed by a small rotation / translation
        dR = bs.qmat(bs.qexp(rng.randn(3) * 2e-3))
        Rp, tp = np.ascontiguousarray(dR @ R), t + rng.randn(3) * 5e-3
        mask = np.zeros(len(X), np.uint8)
        Ro, to = np.zeros(9), np.zeros(3)
        cnt = gh.gh_pnp_parsac_imu(_p(X), _p(x), _p(lens), len(X), _p(Rp), _p(tp), C.c_double(0.2), C.c_double(1.0 / 458.0),
                                   _p(mask), _p(Ro), _p(to))
        assert cnt == mask.sum()
        assert mask[:n_static].mean() > 0.9, mask[:n_static].mean()
        assert mask[n_static:].mean() < 0.2, mask[n_static:].mean()
        assert np.abs(Ro.reshape(3, 3) - R).max() < 5e-3 and np.abs(to - t).max() < 5e-2


def test_imu_parsac_gives_up_without_support(gh):
    rng = np.random.RandomState(3)
    gh.gh_parsac_reset()
    R, t = _pose(rng)
    X, x = _scene(rng, 60, R, t)
    lens = np.full(60, 5, np.int64)
    R_bad, t_bad = _pose(rng, 1.0, 2.0)                 # a prediction nothing agrees with
    mask = np.zeros(60, np.uint8)
    Ro, to = np.zeros(9), np.zeros(3)
    cnt = gh.gh_pnp_parsac_imu(_p(X), _p(x), _p(lens), 60, _p(np.ascontiguousarray(R_bad)), _p(t_bad), C.c_double(0.2),
                               C.c_double(1.0 / 458.0), _p(mask), _p(Ro), _p(to))
    assert cnt == 60 and np.allclose(Ro.reshape(3, 3), np.eye(3)) and np.allclose(to, 0)   # "everything is an inlier", identity


def test_essential_parsac_separates_two_motions(gh):
    rng = np.random.RandomState(4)
    gh.gh_parsac_reset()
    R, t = _pose(rng, 0.15, 0.4)
    n_static, n_moving = 100, 25
    X, x2 = _scene(rng, n_static + n_moving, R, t)
    x1 = X[:, :2] / X[:, 2:]                             # first camera at the origin
    keep = X[:, 2] > 0.5
    X, x1, x2 = X[keep], x1[keep], x2[keep]
    n = len(X)
    moving = np.zeros(n, bool)
    moving[-n_moving:] = True
    x2[moving] += np.array([0.04, -0.03])                # independent image motion of the object
    mask = np.zeros(n, np.uint8)
    E = np.zeros(9)
    gh.gh_essential_parsac(_p(np.ascontiguousarray(x1)), _p(np.ascontiguousarray(x2)), n, C.c_double(1.0 / 458.0), _p(mask), _p(E))
    assert mask[~moving].mean() > 0.9
    assert mask[moving].mean() < 0.3
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Et = tx @ R
    Em = E.reshape(3, 3)
    Em = Em / np.linalg.norm(Em) * np.linalg.norm(Et)
    assert min(np.abs(Em - Et).max(), np.abs(Em + Et).max()) < 5e-2
