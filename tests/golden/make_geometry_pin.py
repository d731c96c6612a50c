"""Generates tests/golden/geometry_pin.npz: bit patterns of the host solvers' outputs (5-point essential solver, essential RANSAC,
Jacobi SVD, real eigen-solver) on seeded inputs, through tests/host_check/geom_host.cpp.

    python tests/golden/make_geometry_pin.py [path/to/libgeom_host.so]

The committed file was produced by the round-3 implementation (dense 64-slot polynomials, run-time-sized SVD / eigen loops) built from
the commit before the solver was restructured (round 4: compact polynomials, fixed-size instances, cached column norms);
tests/test_host_geometry.py::test_solver_outputs_are_pinned_bit_for_bit holds every later version to it.  The pipeline's trajectories
depend on these bits (the initialiser decomposes the RANSAC winner), so a change here is a change of every pipeline golden."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def inputs():
    rng = np.random.RandomState(20260925)
    five = []
    for k in range(400):
        th, t = 0.05 * rng.randn(), 0.1 * rng.randn(3)
        X = np.stack([6 * (rng.rand(5) - 0.5), 4 * (rng.rand(5) - 0.5), 2 + 6 * rng.rand(5)], 1)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        Y = X @ R.T + t
        p1, p2 = X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:]
        if k % 10 == 1:
            p2 = p1.copy()                       # no motion
        if k % 10 == 2:
            p2 = rng.randn(5, 2)                 # garbage
        if k % 10 == 3:
            p1[4], p2[4] = p1[0], p2[0]          # repeated point
        if k % 10 == 4:
            p1 = np.round(p1 * 8) / 8            # exact zeros / ties
        five.append((np.ascontiguousarray(p1), np.ascontiguousarray(p2)))
    many = []
    for k in range(40):
        n = 40 + 7 * k
        th, t = 0.02 * rng.randn(), 0.05 * rng.randn(3)
        X = np.stack([6 * (rng.rand(n) - 0.5), 4 * (rng.rand(n) - 0.5), 2 + 6 * rng.rand(n)], 1)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        Y = X @ R.T + t
        p1, p2 = X[:, :2] / X[:, 2:] + 1e-4 * rng.randn(n, 2), Y[:, :2] / Y[:, 2:] + 1e-4 * rng.randn(n, 2)
        p2[::6] += 0.05 * rng.randn(len(p2[::6]), 2)
        many.append((np.ascontiguousarray(p1), np.ascontiguousarray(p2)))
    mats = [np.ascontiguousarray(rng.randn(m, n)) for (m, n) in ((5, 9), (3, 3), (12, 4), (4, 4), (9, 9)) for _ in range(20)]
    sq = [np.ascontiguousarray(rng.randn(n, n)) for n in (10, 10, 10, 6, 4) for _ in range(20)]
    return five, many, mats, sq


def run(lib):
    gh = C.CDLL(lib)
    five, many, mats, sq = inputs()
    out = {}
    E5 = np.zeros((len(five), 10, 9))
    n5 = np.zeros(len(five), np.int32)
    for k, (p1, p2) in enumerate(five):
        Es = np.zeros((10, 9))
        n5[k] = gh.gh_essential_5pt(_p(p1), _p(p2), _p(Es))
        E5[k] = Es
    out["five_count"], out["five_E"] = n5, E5.view(np.uint64)
    ER = np.zeros((len(many), 9))
    masks = []
    for k, (p1, p2) in enumerate(many):
        mask = np.zeros(len(p1), np.int8)
        E9 = np.zeros(9)
        gh.gh_find_essential(_p(p1), _p(p2), len(p1), _p(mask), _p(E9))
        ER[k] = E9
        masks.append(mask)
    out["ransac_E"], out["ransac_mask"] = ER.view(np.uint64), np.concatenate(masks)
    sv = []
    for A in mats:
        m, n = A.shape
        s, V = np.zeros(n), np.zeros((n, n))
        gh.gh_svd(_p(A), m, n, _p(s), _p(V))
        sv += [s, V.ravel()]
    out["svd"] = np.concatenate(sv).view(np.uint64)
    ev = []
    for M in sq:
        n = len(M)
        wr, wi, V = np.zeros(n), np.zeros(n), np.zeros((n, n))
        gh.gh_real_eigen(_p(M), n, _p(wr), _p(wi), _p(V))
        ev += [wr, wi, V.ravel()]
    out["eigen"] = np.concatenate(ev).view(np.uint64)
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "host_check", "_build", "libgeom_host.so")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "geometry_pin.npz"), **run(lib))
    print("wrote tests/golden/geometry_pin.npz from", lib)
