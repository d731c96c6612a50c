"""CPU tests of the initialiser's two-view geometry and small dense solves (xrslam_amd/csrc/host/two_view.hpp)
against ground-truth geometry and numpy.  The reference versions need Eigen (not in this image), so these checks
-- not reference outputs -- are what pins this code ("parity unpinned", see the header of two_view.hpp)."""
import ctypes as C

import numpy as np
import pytest

from tests import ba_synth as bs
from tests.test_host_geometry import gh  # noqa: F401  (fixture: builds tests/host_check/geom_host.cpp)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rot(rng, angle):
    ax = rng.randn(3)
    ax /= np.linalg.norm(ax)
    return bs.qmat(bs.qexp(ax * angle))


def _same_rotation(A, B, tol=1e-8):
    return np.abs(A - B).max() < tol


def test_homography_4pt_exact(gh):
    rng = np.random.RandomState(3)
    for _ in range(20):
        H = np.eye(3) + 0.3 * rng.randn(3, 3)
        a = rng.rand(4, 2) - 0.5
        b = (np.c_[a, np.ones(4)] @ H.T)
        b = b[:, :2] / b[:, 2:]
        out = np.zeros(9)
        gh.gh_homography_4pt(_p(a), _p(b), _p(out))
        G = out.reshape(3, 3)
        G = G / G[2, 2] * H[2, 2]
        assert np.abs(G - H).max() < 1e-8


def _plane_scene(rng, n, R, t, normal, d):
    """points on the plane normal . X = d in camera 1; camera 2 sees X2 = R X1 + t"""
    X = []
    while len(X) < n:
        ray = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4), 1.0])
        lam = d / (normal @ ray)
        if lam > 0.5:
            X.append(ray * lam)
    X = np.array(X)
    Y = X @ R.T + t
    return X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:], X


def test_homography_ransac_and_decomposition(gh):
    rng = np.random.RandomState(5)
    for trial in range(10):
        R = _rot(rng, 0.2)
        t = rng.randn(3) * 0.3
        normal = np.array([0.1 * rng.randn(), 0.1 * rng.randn(), 1.0])
        normal /= np.linalg.norm(normal)
        d = 4.0
        a, b, _ = _plane_scene(rng, 120, R, t, normal, d)
        b_noisy = b.copy()
        out_idx = rng.choice(120, 25, replace=False)
        b_noisy[out_idx] += rng.uniform(0.05, 0.2, (25, 2))
        mask = np.zeros(120, np.uint8)
        H = np.zeros(9)
        cnt = gh.gh_find_homography(_p(a), _p(b_noisy), 120, C.c_double(0.7 / 458.0), 648, _p(mask), _p(H))
        assert cnt >= 90 and mask[out_idx].sum() <= 2
        Ht = R + np.outer(t, normal) / d
        G = H.reshape(3, 3)
        G = G / np.linalg.svd(G)[1][1] * np.sign(np.linalg.det(G))
        Ht = Ht / np.linalg.svd(Ht)[1][1]
        assert np.abs(G - Ht).max() < 5e-3
        # decomposition of the exact homography holds the true (R, t/|t|, n) as one of its two solutions
        Rs, Ts, ns = np.zeros(18), np.zeros(6), np.zeros(6)
        for sign in (1.0, -1.0):   # the sign of H must not matter
            Hin = np.ascontiguousarray(sign * 2.5 * (R + np.outer(t, normal) / d))
            assert gh.gh_decompose_homography(_p(Hin), _p(Rs), _p(Ts), _p(ns)) == 1
            hit = False
            for k in range(2):
                Rk, Tk, nk = Rs[9 * k:9 * k + 9].reshape(3, 3), Ts[3 * k:3 * k + 3], ns[3 * k:3 * k + 3]
                assert abs(np.linalg.det(Rk) - 1) < 1e-9 and np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-9
                if _same_rotation(Rk, R, 1e-7):
                    s = np.sign(nk @ normal)
                    assert np.abs(s * nk - normal).max() < 1e-7
                    assert np.abs(s * Tk / np.linalg.norm(Tk) - t / np.linalg.norm(t)).max() < 1e-7
                    hit = True
            assert hit


def test_pure_rotation_is_refused(gh):
    rng = np.random.RandomState(7)
    R = _rot(rng, 0.3)
    Rs, Ts, ns = np.zeros(18), np.zeros(6), np.zeros(6)
    assert gh.gh_decompose_homography(_p(np.ascontiguousarray(1.7 * R)), _p(Rs), _p(Ts), _p(ns)) == 0


def test_essential_decomposition(gh):
    rng = np.random.RandomState(11)
    for _ in range(20):
        R = _rot(rng, rng.uniform(0.05, 1.0))
        t = rng.randn(3)
        t /= np.linalg.norm(t)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = np.ascontiguousarray((tx @ R) * rng.choice([-1.0, 1.0]) * rng.uniform(0.5, 3))
        Rs, T = np.zeros(18), np.zeros(3)
        gh.gh_decompose_essential(_p(E), _p(Rs), _p(T))
        R1, R2 = Rs[:9].reshape(3, 3), Rs[9:].reshape(3, 3)
        for Rk in (R1, R2):
            assert abs(np.linalg.det(Rk) - 1) < 1e-9 and np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-9
        assert _same_rotation(R1, R, 1e-7) or _same_rotation(R2, R, 1e-7)
        assert min(np.abs(T - t).max(), np.abs(T + t).max()) < 1e-7


def test_least_squares_matches_lapack(gh):
    rng = np.random.RandomState(13)
    for m, n in ((42, 28), (42, 27), (12, 12), (30, 5)):
        A = rng.randn(m, n)
        A[:, rng.randint(n)] *= 1e-3
        b = rng.randn(m)
        x = np.zeros(n)
        gh.gh_lstsq(_p(A), _p(b), m, n, _p(x))
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.abs(x - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    # rank deficient: still a minimiser of the residual
    A = rng.randn(20, 6)
    A[:, 5] = A[:, 0] + A[:, 1]
    b = rng.randn(20)
    x = np.zeros(6)
    gh.gh_lstsq(_p(A), _p(b), 20, 6, _p(x))
    ref = np.linalg.lstsq(A, b, rcond=None)[0]
    assert abs(np.linalg.norm(A @ x - b) - np.linalg.norm(A @ ref - b)) < 1e-10


def test_small_solves_and_rotations(gh):
    rng = np.random.RandomState(17)
    for _ in range(20):
        J = rng.randn(7, 3)
        A = np.ascontiguousarray(J.T @ J)
        b = rng.randn(3)
        x = np.zeros(3)
        gh.gh_svd_solve3(_p(A), _p(b), _p(x))
        assert np.abs(x - np.linalg.solve(A, b)).max() < 1e-9
        R = np.ascontiguousarray(_rot(rng, rng.uniform(0, np.pi)))
        q = np.zeros(4)
        gh.gh_quat_from_matrix(_p(R), _p(q))
        assert abs(np.linalg.norm(q) - 1) < 1e-12 and np.abs(bs.qmat(q) - R).max() < 1e-12
        w = np.zeros(3)
        gh.gh_logmap(_p(q), _p(w))
        assert np.abs(bs.qmat(bs.qexp(w)) - R).max() < 1e-9 and np.linalg.norm(w) <= np.pi + 1e-12
        a, c = rng.randn(3), rng.randn(3)
        gh.gh_from_two_vectors(_p(a), _p(c), _p(q))
        assert np.abs(bs.qmat(q) @ (a / np.linalg.norm(a)) - c / np.linalg.norm(c)).max() < 1e-12
        basis = np.zeros(6)
        gh.gh_s2_basis(_p(a), _p(basis))
        b1, b2 = basis[:3], basis[3:]
        assert abs(b1 @ a) < 1e-12 and abs(b2 @ a) < 1e-12 and abs(b1 @ b2) < 1e-12
        assert abs(np.linalg.norm(b1) - 1) < 1e-12 and abs(np.linalg.norm(b2) - 1) < 1e-12
    # singular normal matrix: minimum-norm solution like the pseudo-inverse
    J = rng.randn(5, 3)
    J[:, 2] = 0
    A = np.ascontiguousarray(J.T @ J)
    b = np.ascontiguousarray(J.T @ rng.randn(5))
    x = np.zeros(3)
    gh.gh_svd_solve3(_p(A), _p(b), _p(x))
    assert np.abs(x - np.linalg.pinv(A) @ b).max() < 1e-9
    # opposite vectors: a half turn about a perpendicular axis
    a = np.array([0.3, -0.2, 0.9])
    q = np.zeros(4)
    gh.gh_from_two_vectors(_p(a), _p(np.ascontiguousarray(-a)), _p(q))
    assert np.abs(bs.qmat(q) @ a + a).max() < 1e-6
