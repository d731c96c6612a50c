"""CPU tests of the product's host-side geometry / config code (xrslam_amd/csrc/host) against numpy.
These pieces stay on the host in the product (sequential, order-defining, a handful of points per frame);
Eigen is replaced by hand-written SVD / eigen-solvers which are checked here against LAPACK."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import ba_synth as bs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_check", "geom_host.cpp")
OUT = os.path.join(ROOT, "tests", "host_check", "_build", "libgeom_host.so")


@pytest.fixture(scope="module")
def gh():
    deps = [SRC] + [os.path.join(ROOT, "xrslam_amd", "csrc", "host", f) for f in ("hla.hpp", "geometry.hpp", "config.hpp", "two_view.hpp", "epnp.hpp", "parsac.hpp")]
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", OUT])
    return C.CDLL(OUT)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_real_eigen_matches_lapack(gh):
    rng = np.random.RandomState(0)
    for trial in range(20):
        n = 10
        M = rng.randn(n, n)
        if trial % 2:
            M[6:] = 0                      # companion-like structure of the 5-point action matrix
            M[6, 0] = M[7, 1] = M[8, 3] = M[9, 6] = 1
        wr = np.zeros(n); wi = np.zeros(n); V = np.zeros((n, n))
        gh.gh_real_eigen(_p(np.ascontiguousarray(M)), n, _p(wr), _p(wi), _p(V))
        ref = np.linalg.eigvals(M)
        got = wr + 1j * wi
        assert np.abs(np.sort_complex(got) - np.sort_complex(ref)).max() < 1e-8 * max(1, np.abs(ref).max())
        for e in range(n):
            if abs(wi[e]) < 1e-10:
                v = V[:, e]
                assert np.linalg.norm(v) > 0.99
                assert np.linalg.norm(M @ v - wr[e] * v) < 1e-6 * max(1, np.abs(ref).max())


def test_svd_nullspace_and_values(gh):
    rng = np.random.RandomState(1)
    for (m, n) in ((5, 9), (3, 3), (12, 4), (20, 4)):
        A = rng.randn(m, n)
        s = np.zeros(n); V = np.zeros((n, n))
        gh.gh_svd(_p(np.ascontiguousarray(A)), m, n, _p(s), _p(V))
        ref = np.linalg.svd(A, compute_uv=False)
        np.testing.assert_allclose(s[:len(ref)], ref, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)
        if m < n:
            assert np.abs(A @ V[:, m:]).max() < 1e-12


def _two_view(rng, n, noise=0.0):
    R = bs.qmat(bs.qexp(rng.randn(3) * 0.1))
    t = rng.randn(3)
    t /= np.linalg.norm(t)
    X = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(3, 8, n)])
    X2 = (R @ X.T).T + t
    p1 = X[:, :2] / X[:, 2:] + rng.randn(n, 2) * noise
    p2 = X2[:, :2] / X2[:, 2:] + rng.randn(n, 2) * noise
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return p1, p2, tx @ R, R, X, X2


def test_five_point_contains_true_essential(gh):
    rng = np.random.RandomState(2)
    hits = 0
    for _ in range(20):
        p1, p2, E_true, _, _, _ = _two_view(rng, 5)
        Es = np.zeros((10, 9))
        k = gh.gh_essential_5pt(_p(np.ascontiguousarray(p1)), _p(np.ascontiguousarray(p2)), _p(Es))
        assert 1 <= k <= 10
        best = 1e9
        for e in Es[:k].reshape(-1, 3, 3):
            h1 = np.column_stack([p1, np.ones(5)]); h2 = np.column_stack([p2, np.ones(5)])
            assert np.abs(np.einsum("ni,ij,nj->n", h2, e, h1)).max() < 1e-8 * np.abs(e).max()   # epipolar constraints
            en = e / np.linalg.norm(e)
            tn = E_true / np.linalg.norm(E_true)
            best = min(best, np.linalg.norm(en - tn), np.linalg.norm(en + tn))
            c = 2 * e @ e.T @ e - np.trace(e @ e.T) * e                                           # cubic constraints
            assert np.abs(c).max() < 1e-5 * np.abs(e).max() ** 3 and abs(np.linalg.det(e)) < 1e-5 * np.abs(e).max() ** 3
        hits += best < 1e-6
    assert hits >= 19


def test_rotation_2pt_and_ransac(gh):
    rng = np.random.RandomState(3)
    R = bs.qmat(bs.qexp(rng.randn(3) * 0.3))
    a = rng.randn(40, 3); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = (R @ a.T).T
    R9 = np.zeros(9)
    gh.gh_rotation_2pt(_p(np.ascontiguousarray(a[:2])), _p(np.ascontiguousarray(b[:2])), _p(R9))
    np.testing.assert_allclose(R9.reshape(3, 3), R, atol=1e-10)
    b += rng.randn(40, 3) * 1e-4                   # exact data would give acos(1 + eps) = NaN -> outlier (as in the reference)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[5] = -b[5]                                   # an outlier
    mask = np.zeros(40, np.int8)
    n = gh.gh_find_rotation(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), 40, C.c_double(np.pi / 18), _p(mask), _p(R9))
    assert n == 40 and mask[5] == 0 and mask.sum() == 39
    np.testing.assert_allclose(R9.reshape(3, 3), R, atol=1e-3)


def test_essential_ransac_mask(gh):
    rng = np.random.RandomState(4)
    p1, p2, E_true, *_ = _two_view(rng, 120, noise=1e-3)
    mask = np.zeros(120, np.int8); E9 = np.zeros(9)
    n = gh.gh_find_essential(_p(np.ascontiguousarray(p1)), _p(np.ascontiguousarray(p2)), 120, _p(mask), _p(E9))
    assert n == 120 and mask.all()                 # threshold 7.68 in normalised units accepts everything finite
    p2[7] = np.nan
    n = gh.gh_find_essential(_p(np.ascontiguousarray(p1)), _p(np.ascontiguousarray(p2)), 120, _p(mask), _p(E9))
    assert mask[7] == 0 and mask.sum() == 119
    n = gh.gh_find_essential(_p(np.ascontiguousarray(p1[:3])), _p(np.ascontiguousarray(p2[:3])), 3, _p(mask), _p(E9))
    assert n == 3 and not mask[:3].any()           # fewer than 5 points: all outliers (ransac.h:41-45)


def test_triangulation(gh):
    rng = np.random.RandomState(5)
    X = np.array([0.3, -0.2, 5.0])
    Ps, zs = [], []
    for k in range(4):
        R = bs.qmat(bs.qexp(rng.randn(3) * 0.05)); t = rng.randn(3) * 0.3
        Ps.append(np.column_stack([R, t]))
        y = R @ X + t
        zs.append(y / np.linalg.norm(y))
    h = np.zeros(4)
    gh.gh_triangulate(_p(np.ascontiguousarray(np.array(Ps))), _p(np.ascontiguousarray(np.array(zs))), 4, _p(h))
    np.testing.assert_allclose(h[:3] / h[3], X, atol=1e-9)


def test_lotbox_is_libstdcxx_minstd_sequence(gh):
    """LotBox draws from std::default_random_engine (minstd_rand0, a=16807) through
    std::uniform_int_distribution<size_t>; restated here independently (libstdc++ downscaling rule)."""
    size, rounds, draws = 150, 6, 5
    out = np.zeros(rounds * draws, np.int64)
    gh.gh_lotbox(size, 0, rounds, draws, _p(out))
    state = 1                                        # seed(0) -> 1 for a multiplicative LCG

    def urng():
        nonlocal state
        state = state * 16807 % 2147483647
        return state

    def uniform(lo, hi):
        urange, urngrange = hi - lo, 2147483646 - 1
        uerange = urange + 1
        scaling = urngrange // uerange
        past = uerange * scaling
        while True:
            r = urng() - 1
            if r < past:
                return lo + r // scaling

    exp = []
    for _ in range(rounds):
        lots = list(range(size)) if not exp else lots
        cap = 0
        for _ in range(draws):
            j = uniform(cap, size - 1)
            lots[cap], lots[j] = lots[j], lots[cap]
            exp.append(lots[cap])
            cap += 1
    np.testing.assert_array_equal(out, exp)


def test_yaml_config_surface(gh, tmp_path):
    out = np.zeros(64)
    rc = gh.gh_load_config(os.path.join(ROOT, "configs", "euroc_slam.yaml").encode(),
                           os.path.join(ROOT, "configs", "euroc_sensor.yaml").encode(), _p(out))
    assert rc == 0
    np.testing.assert_allclose(out[:6], [752, 480, 458.654, 457.296, 367.215, 248.375])
    np.testing.assert_allclose(out[6:10], bs.Q_BC / np.linalg.norm(bs.Q_BC), atol=1e-15)
    np.testing.assert_allclose(out[10:13], bs.P_BC)
    np.testing.assert_allclose(out[13:17], [2.8791302399999997e-08, 4.0e-6, 3.7608844899999997e-10, 9.0e-6])
    assert list(out[17:21]) == [10, 200, 20, 30] and out[21] == 0.02 and out[22] == 1 and out[23] == 0.5
    # defaults when a slam key is absent (config.cpp:7-78), errors when a device key is missing
    slam = tmp_path / "slam.yaml"
    slam.write_text("%YAML:1.0\nsolver:\n  iteration_limit: 7\n")
    rc = gh.gh_load_config(str(slam).encode(), os.path.join(ROOT, "configs", "euroc_sensor.yaml").encode(), _p(out))
    assert rc == 0 and out[17] == 10 and out[18] == 150 and out[19] == 200 and out[20] == 7 and out[21] == 0.1
    dev = tmp_path / "dev.yaml"
    dev.write_text("%YAML:1.0\ncam0:\n  resolution: [752, 480]\n")
    assert gh.gh_load_config(str(slam).encode(), str(dev).encode(), _p(out)) == -1


def test_poisson_disk_dense_scan_equals_the_cell_by_cell_scan(gh):
    """PoissonDisk2::permit reads the scan box row by row when it lies inside the dense cell array (every point the tracker
    produces) and cell by cell, with the hash fallback, otherwise: same verdict on every point of random sequences, including
    points on the image border, outside it, and at spacings around the radius."""
    import ctypes as C
    gh.gh_poisson_disagreements.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.RandomState(11)
    for radius, w, h, n in ((20.0, 752, 480, 1500), (7.5, 320, 240, 3000), (33.0, 1280, 720, 1200)):
        xy = np.c_[rng.uniform(-40, w + 40, n), rng.uniform(-40, h + 40, n)]
        xy[: n // 3] = np.clip(xy[: n // 3], 0, [w - 1, h - 1])                    # a third strictly inside
        xy[n // 3: n // 2] = np.round(xy[n // 3: n // 2])                          # integer pixels (the detector's output)
        xy = np.ascontiguousarray(xy)
        acc = np.zeros(n, np.uint8)
        assert gh.gh_poisson_disagreements(xy.ctypes.data, n, radius, w, h, acc.ctypes.data) == 0
        assert 0 < acc.sum() < n                                                   # the filter both accepts and refuses here


def test_five_point_returns_every_real_essential_matrix(gh):
    """Completeness of solve_essential_5pt (geometry/essential.cpp:105-297), independent of its action-matrix / eigen-solver route:
    a numpy multi-start Gauss-Newton search over the null space E(x, y, z) = x E1 + y E2 + z E3 + E4 of the five epipolar
    constraints (numpy's own SVD basis) collects the real roots of det E = 0, 2 E E^T E - tr(E E^T) E = 0; every root it finds is
    among the matrices the solver returned (up to scale and sign), and the solver returned nothing that is not a root."""
    rng = np.random.RandomState(8)

    def unit(e):
        e = e / np.linalg.norm(e)
        return e * np.sign(e.flat[np.argmax(np.abs(e))])

    def residual(e):
        return np.concatenate([(2 * e @ e.T @ e - np.trace(e @ e.T) * e).ravel(), [np.linalg.det(e)]])

    total_cpp = total_np = 0
    for trial in range(8):
        p1, p2, _E, _, _, _ = _two_view(rng, 5, noise=0.0 if trial % 2 else 0.01)     # (noise: no exact geometry behind the points)
        Es = np.zeros((10, 9))
        k = gh.gh_essential_5pt(_p(np.ascontiguousarray(p1)), _p(np.ascontiguousarray(p2)), _p(Es))
        cpp = [unit(e) for e in Es[:k].reshape(-1, 3, 3)]
        # the solver's own output: every matrix is a root
        for e in cpp:
            assert np.abs(residual(e)).max() < 1e-7
        # independent search
        h1 = np.column_stack([p1, np.ones(5)]); h2 = np.column_stack([p2, np.ones(5)])
        A = np.stack([np.outer(h2[i], h1[i]).ravel() for i in range(5)])               # h2^T E h1 = 0, E row-major
        B = np.linalg.svd(A)[2][5:].reshape(4, 3, 3)
        roots = []
        for start in range(480):
            v = rng.randn(3) * (0.3, 1.0, 3.0, 10.0)[start % 4]
            for _ in range(60):
                e = v[0] * B[0] + v[1] * B[1] + v[2] * B[2] + B[3]
                f = residual(e)
                J = np.empty((10, 3))
                for c in range(3):
                    dv = np.zeros(3); dv[c] = 1e-7
                    J[:, c] = (residual(e + dv[c] * B[c]) - f) / 1e-7
                step = np.linalg.lstsq(J, -f, rcond=None)[0]
                v = v + step
                if np.linalg.norm(step) < 1e-13 * max(1.0, np.linalg.norm(v)):
                    break
            e = v[0] * B[0] + v[1] * B[1] + v[2] * B[2] + B[3]
            if np.abs(residual(unit(e))).max() < 1e-9 and np.linalg.norm(v) < 1e6:
                u = unit(e)
                if not any(np.linalg.norm(u - r) < 1e-6 for r in roots):
                    roots.append(u)
        assert roots, trial
        for u in roots:
            assert min(np.linalg.norm(u - c) for c in cpp) < 1e-6, "trial %d: a real root the solver did not return" % trial
        total_cpp += len(cpp)
        total_np += len(roots)
    assert total_np >= 0.75 * total_cpp, (total_np, total_cpp)     # (the search is stochastic: it need not find every root, only no extra one)


def test_solver_outputs_are_pinned_bit_for_bit(gh):
    """The 5-point solver, the essential RANSAC, the Jacobi SVD and the eigen-solver return exactly the bits of the committed fixture
    (tests/golden/geometry_pin.npz, produced by the round-3 implementation: tests/golden/make_geometry_pin.py).  Round 4 restructured
    them for speed (compact polynomials, fixed-size instances, cached column norms) under the contract that no operation and no
    order of operations changes: the pipeline's trajectories and every pipeline golden depend on these bits."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_geometry_pin", os.path.join(ROOT, "tests", "golden", "make_geometry_pin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.run(OUT)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "geometry_pin.npz"))
    assert set(got) == set(ref.files)
    for k in ref.files:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert ref["five_count"].sum() > 1000 and ref["ransac_mask"].sum() > 1000   # the fixture is not vacuous
