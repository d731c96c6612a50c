"""CPU tests of the KLT oracle (oracle/klt_oracle.c): golden vectors of the
reference's own test, independent numpy cross-checks, and domain properties."""
import numpy as np
import pytest

from oracle import klt_oracle as ko
from tests.util import noise_image, warp_affine


def test_reference_known_answers(golden_pair, klt_expected):
    """xrslam-test/test/src/test_feature_track.cpp:41,64 pins 164 detected / 161 tracked (counts only, unrecorded
    OpenCV build).  The restatement reproduces both exactly once the undistortion map passes through float32 before
    its 1/32-pixel rounding (oracle/undistort.py; rounding straight from the double changes 36 pixels and gives
    165 / 162 -- the *_v1 fixtures).  The exact oracle outputs are pinned as a regression too."""
    a, b = golden_pair
    A, B = ko.OracleImage(a), ko.OracleImage(b)
    A.preprocess(6.0, 8, 8)
    B.preprocess(6.0, 8, 8)
    kp = A.detect_keypoints(np.zeros((0, 2)), 200, 20.0)
    assert len(kp) == 164
    nx, st = A.track_keypoints(B, kp, kp.copy())
    assert int(st.sum()) == 161
    np.testing.assert_array_equal(kp, klt_expected["keypoints"])
    np.testing.assert_array_equal(st, klt_expected["status"])
    np.testing.assert_array_equal(nx, klt_expected["next"])


def test_reference_known_answers_at_frame_level(golden_pair, golden_pair_v1):
    """The reference's test_feature_track end to end (test_feature_track.cpp:24-65) through the product's host
    pipeline linked against the oracle (tests/host_check/frame_host.cpp): preprocess + Frame::detect_keypoints on
    frame 1, Frame::track_keypoints onto frame 2 -- LK both ways, 5-point and 2-point RANSAC gates, Poisson thinning,
    track creation -- with the shipped euroc configuration: keypoint_num() == 164, !FT_NO_TRANSLATION, 161 tracks."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ob = os.path.join(root, "oracle", "_build")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    src = os.path.join(root, "tests", "host_check", "frame_host.cpp")
    out = os.path.join(root, "tests", "host_check", "_build", "libframe_host.so")
    deps = [src] + [os.path.join(ob, f) for f in ("klt_oracle.o", "ba_oracle.o", "xrhip_shim.o")] + \
           [os.path.join(root, "xrslam_amd", "csrc", "host", f) for f in ("pipeline.hpp", "map.hpp", "geometry.hpp", "config.hpp")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-I" + os.path.join(root, "include"), src,
                               os.path.join(ob, "klt_oracle.o"), os.path.join(ob, "ba_oracle.o"), os.path.join(ob, "xrhip_shim.o"),
                               "-o", out, "-lm"])
    lib = C.CDLL(out)

    def run(pair):
        a, b = [np.ascontiguousarray(x) for x in pair]
        res = (C.c_int * 3)()
        rc = lib.fh_feature_track(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.shape[1],
                                  os.path.join(root, "configs", "euroc_slam.yaml").encode(),
                                  os.path.join(root, "configs", "euroc_sensor.yaml").encode(), res)
        assert rc == 0
        return list(res)

    assert run(golden_pair) == [164, 0, 161]          # the reference's three assertions
    assert run(golden_pair_v1) == [165, 0, 162]       # the double-rounded undistortion map is one corner off


def _clahe_numpy(g, clip_limit=6.0, tiles=8):
    """Independent numpy restatement of cv::CLAHE (vectorised, float32)."""
    h, w = g.shape
    tw, th = w // tiles, h // tiles
    area = tw * th
    clip = max(int(clip_limit * area / 256), 1)
    lut = np.zeros((tiles, tiles, 256), np.float32)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(g[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            ex = np.maximum(hist - clip, 0).sum()
            hist = np.minimum(hist, clip)
            rb = ex // 256
            res = ex - rb * 256
            hist += rb
            if res:
                step = max(256 // res, 1)
                hist[np.arange(0, 256, step)[:res]] += 1
            cs = np.cumsum(hist).astype(np.float32)
            lut[ty, tx] = np.clip(np.rint(cs * (np.float32(255.0) / np.float32(area))), 0, 255)
    x = np.arange(w, dtype=np.float32)
    y = np.arange(h, dtype=np.float32)
    txf = x * (np.float32(1) / np.float32(tw)) - np.float32(0.5)
    tyf = y * (np.float32(1) / np.float32(th)) - np.float32(0.5)
    tx1 = np.floor(txf).astype(int)
    ty1 = np.floor(tyf).astype(int)
    xa = (txf - tx1.astype(np.float32))[None, :]
    ya = (tyf - ty1.astype(np.float32))[:, None]
    tx2 = np.minimum(tx1 + 1, tiles - 1)[None, :]
    ty2 = np.minimum(ty1 + 1, tiles - 1)[:, None]
    tx1 = np.maximum(tx1, 0)[None, :]
    ty1 = np.maximum(ty1, 0)[:, None]
    xa1 = np.float32(1) - xa
    ya1 = np.float32(1) - ya
    r = (lut[ty1, tx1, g] * xa1 + lut[ty1, tx2, g] * xa) * ya1 + (lut[ty2, tx1, g] * xa1 + lut[ty2, tx2, g] * xa) * ya
    return np.clip(np.rint(r), 0, 255).astype(np.uint8)


def test_clahe_matches_independent_numpy(golden_pair):
    a, _ = golden_pair
    np.testing.assert_array_equal(ko.clahe(a), _clahe_numpy(a))
    s = noise_image(640, 480, seed=3)
    np.testing.assert_array_equal(ko.clahe(s), _clahe_numpy(s))


def test_pyramid_numpy_crosscheck():
    """pyrDown and Scharr against straightforward numpy with reflect-101 padding."""
    g = noise_image(752, 480, seed=5)
    A = ko.OracleImage(g)
    A.preprocess(6.0, 8, 8)
    prev = A.image.astype(np.int64)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    for l in range(4):
        img, der = A.level(l)
        if l > 0:
            p = np.pad(prev, 2, mode="reflect")
            rows = sum(k[i] * p[:, i:i + prev.shape[1]] for i in range(5))
            full = sum(k[j] * rows[j:j + prev.shape[0], :] for j in range(5))
            exp = ((full[::2, ::2] + 128) >> 8)
            np.testing.assert_array_equal(img, exp.astype(np.uint8))
        else:
            np.testing.assert_array_equal(img, A.image)
        I = np.pad(img.astype(np.int64), 1, mode="reflect")
        h, w = img.shape
        t0 = (I[0:h, :] + I[2:h + 2, :]) * 3 + I[1:h + 1, :] * 10
        t1 = I[2:h + 2, :] - I[0:h, :]
        dx = t0[:, 2:] - t0[:, :-2]
        dy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
        np.testing.assert_array_equal(der[..., 0], dx.astype(np.int16))
        np.testing.assert_array_equal(der[..., 1], dy.astype(np.int16))
        pimg, pder = A.level(l, padded=True)
        np.testing.assert_array_equal(pimg, np.pad(img, 21, mode="reflect"))
        np.testing.assert_array_equal(pder[21:-21, 21:-21], der)
        assert not pder[:21].any() and not pder[:, :21].any() and not pder[-21:].any() and not pder[:, -21:].any()
        prev = img.astype(np.int64)


def test_lk_recovers_known_translation():
    g = noise_image(752, 480, seed=7)
    shift = np.array([3.25, -2.5])
    g2 = warp_affine(g, np.eye(2), -shift)      # g2(x) = g(x - shift): content moves by +shift
    A, B = ko.OracleImage(g), ko.OracleImage(g2)
    A.preprocess()
    B.preprocess()
    kp = A.detect_keypoints(np.zeros((0, 2)), 150, 20.0)
    assert len(kp) > 100
    nx, st = A.track_keypoints(B, kp, None)
    good = st > 0
    assert good.mean() > 0.9
    err = np.abs(nx[good] - kp[good] - shift)
    assert np.median(err) < 0.05 and err.max() < 0.5


def test_lk_identity_and_flat():
    g = noise_image(640, 480, seed=9)
    A, B = ko.OracleImage(g), ko.OracleImage(g.copy())
    A.preprocess()
    B.preprocess()
    kp = A.detect_keypoints(np.zeros((0, 2)), 100, 20.0)
    nx, st = A.track_keypoints(B, kp, kp.copy())
    assert st.all()
    np.testing.assert_allclose(nx, kp, atol=1e-3)
    flat = np.full((480, 640), 100, np.uint8)
    F, G = ko.OracleImage(flat), ko.OracleImage(flat.copy())
    F.preprocess()
    G.preprocess()
    pts = np.array([[100.0, 100.0], [320.5, 240.25]])
    _, st = F.track_keypoints(G, pts, pts.copy())
    assert not st.any()          # minEig test must reject textureless windows


def test_track_empty_and_border():
    g = noise_image(640, 480, seed=11)
    A, B = ko.OracleImage(g), ko.OracleImage(g.copy())
    A.preprocess()
    B.preprocess()
    nx, st = A.track_keypoints(B, np.zeros((0, 2)), None)
    assert len(st) == 0
    pts = np.array([[5.0, 5.0], [635.0, 475.0], [19.9, 240.0], [320.0, 240.0], [-30.0, 100.0], [700.0, 100.0]])
    nx, st = A.track_keypoints(B, pts, pts.copy())
    assert list(st[:3]) == [0, 0, 0] and st[3] == 1 and st[4] == 0 and st[5] == 0


def test_detect_respects_existing_points_and_border(golden_pair):
    a, _ = golden_pair
    A = ko.OracleImage(a)
    A.preprocess()
    fresh = A.detect_keypoints(np.zeros((0, 2)), 200, 20.0)
    assert (fresh[:, 0] >= 20).all() and (fresh[:, 0] < 752 - 20).all()
    assert (fresh[:, 1] >= 20).all() and (fresh[:, 1] < 480 - 20).all()
    d = np.linalg.norm(fresh[:, None] - fresh[None], axis=2) + np.eye(len(fresh)) * 1e9
    assert d.min() >= 20.0
    again = A.detect_keypoints(fresh[:50], 200, 20.0)
    np.testing.assert_array_equal(again[:50], fresh[:50])
    new = again[50:]
    dd = np.linalg.norm(new[:, None] - fresh[None, :50], axis=2)
    assert dd.min() >= 20.0


def test_poisson_select_matches_bruteforce_semantics():
    rng = np.random.RandomState(0)
    pts = rng.rand(400, 2) * [752, 480]
    keep = ko.poisson_select(pts, 20.0)
    kept = pts[keep > 0]
    d = np.linalg.norm(kept[:, None] - kept[None], axis=2) + np.eye(len(kept)) * 1e9
    # the hash grid only remembers the last point per cell, so the filter is a (slightly
    # permissive) approximation of a strict 20 px exclusion -- exactly like the reference
    assert d.min() > 20.0 / np.sqrt(2.0) - 1e-9
    assert keep[0] == 1


def test_results_do_not_depend_on_thread_count(golden_pair):
    """The oracle's image / point loops may run on several OpenMP threads (bench.py's cpu_baseline_mt leg):
    every iteration is independent, so planes, corners, positions and LK counters are identical."""
    a, b = golden_pair

    def run():
        A, B = ko.OracleImage(a), ko.OracleImage(b)
        A.preprocess(6.0, 8, 8)
        B.preprocess(6.0, 8, 8)
        kp = A.detect_keypoints(np.zeros((0, 2)), 200, 20.0)
        st = ko.LkStats()
        nx, status = A.track_keypoints(B, kp, kp.copy(), stats=st)
        planes = [A.level(l, padded=True) for l in range(4)]
        return kp, nx, status, (st.iters, st.templates), planes

    assert ko.get_threads() == 1
    one = run()
    try:
        ko.set_threads(4)
        four = run()
    finally:
        ko.set_threads(1)
    for x, y in zip(one[:3], four[:3]):
        np.testing.assert_array_equal(x, y)
    assert one[3] == four[3]
    for (i1, d1), (i4, d4) in zip(one[4], four[4]):
        np.testing.assert_array_equal(i1, i4)
        np.testing.assert_array_equal(d1, d4)


def test_float_accumulation_orders_do_not_move_the_known_answers(golden_pair):
    """Deviation #1: the oracle and the HIP kernel sum the LK window terms as exact integers; OpenCV
    accumulates them in float, in an order that depends on its build (scalar: pixel after pixel; SIMD: four strided lanes).
    On the reference's own two frames neither float order flips a status bit -- the known answer 161 stands for all three
    accumulations -- and tracked positions move by less than 1e-3 px (1e-4 relative is north_star's bound on ~400 px
    coordinates: 4e-2 px).  tools/lk_accumulation_study.py prints the same for synthetic pairs."""
    from oracle import klt_oracle as ko
    a, b = golden_pair
    A, B = ko.OracleImage(a), ko.OracleImage(b)
    A.preprocess()
    B.preprocess()
    kp = A.detect_keypoints(np.zeros((0, 2)), 200, 20.0)
    res = {}
    try:
        for mode in (0, 1, 2):
            ko.lib().orc_set_lk_accumulation(mode)
            res[mode] = A.track_keypoints(B, kp, None)
    finally:
        ko.lib().orc_set_lk_accumulation(0)
    p0, s0 = res[0]
    assert len(kp) == 164 and int(s0.sum()) == 161
    for mode in (1, 2):
        p, s = res[mode]
        np.testing.assert_array_equal(s, s0)
        assert np.abs(p[s0 == 1] - p0[s0 == 1]).max() < 1e-3
