"""GPU parity tests of the KLT front-end: HIP path (through the C ABI) vs the CPU
oracle on identical inputs.  Integer/byte data and status bits must be bit-exact;
LK positions are bit-exact too because both sides use exact integer reductions
and identical scalar float math (tolerance kept at 1e-4 relative per north_star,
asserted tighter where it holds)."""
import os

import numpy as np
import pytest

from tests.util import noise_image, warp_affine

pytestmark = pytest.mark.gpu

DUMP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


@pytest.fixture(scope="module")
def mods():
    from oracle import klt_oracle as ko
    from xrslam_amd import klt
    return ko, klt


def _dump(name, **arrs):
    try:
        os.makedirs(DUMP, exist_ok=True)
        np.savez_compressed(os.path.join(DUMP, name + ".npz"), **arrs)
    except Exception:
        pass


def _pair(ko, klt, a, b, clip=6.0, tiles=8, max_points=600):
    h, w = a.shape
    ctx = klt.KltContext(w, h, max_points)
    HA, HB = ctx.image(a), ctx.image(b)
    HA.preprocess(clip, tiles, tiles)
    HB.preprocess(clip, tiles, tiles)
    OA, OB = ko.OracleImage(a), ko.OracleImage(b)
    OA.preprocess(clip, tiles, tiles)
    OB.preprocess(clip, tiles, tiles)
    return ctx, HA, HB, OA, OB


def _assert_pyramid_equal(H, O, tag):
    for l in range(4):
        hi, hd = H.level(l)
        oi, od = O.level(l)
        if not (np.array_equal(hi, oi) and np.array_equal(hd, od)):
            _dump("pyr_mismatch_%s_l%d" % (tag, l), hi=hi, oi=oi, hd=hd, od=od)
        assert hi.shape == oi.shape
        np.testing.assert_array_equal(hi, oi, err_msg="%s level %d image" % (tag, l))
        np.testing.assert_array_equal(hd, od, err_msg="%s level %d deriv" % (tag, l))


@pytest.mark.parametrize("shape,tiles", [((480, 752), 8), ((480, 640), 8), ((479, 641), 8), ((720, 1280), 8),
                                         ((480, 752), 4)])
def test_preprocess_bit_exact(mods, shape, tiles):
    ko, klt = mods
    g = noise_image(shape[1], shape[0], seed=21)
    ctx, HA, HB, OA, OB = _pair(ko, klt, g, g, 6.0, tiles)
    _assert_pyramid_equal(HA, OA, "noise%dx%d" % shape)


@pytest.mark.parametrize("shape,tiles", [((480, 752), 8), ((479, 641), 8), ((720, 1280), 8), ((480, 752), 4), ((352, 353), 8)])
def test_fused_pyramid_matches_the_five_launch_path_border_included(mods, shape, tiles):
    """xrhip_image_preprocess builds the pyramid in one launch (k_pyramid); the five launches it replaces stay behind a switch.
    Same image planes INCLUDING the 21-pixel reflect-101 borders (what the LK windows near the image edge read; the oracle
    comparison above covers the interiors), same derivative planes, at sizes whose levels are odd / not multiples of the tile."""
    ko, klt = mods
    g = noise_image(shape[1], shape[0], seed=77)
    ctx = klt.KltContext(shape[1], shape[0], 300)
    planes = {}
    for fused in (1, 0):
        ctx.set_fused_pyramid(fused)
        im = ctx.image(g)
        im.preprocess(6.0, tiles, tiles)
        planes[fused] = [(im.level_padded(l),) + im.level(l) for l in range(4)]
    ctx.set_fused_pyramid(1)
    for l in range(4):
        for a, b, what in zip(planes[1][l], planes[0][l], ("padded plane", "image", "derivatives")):
            if not np.array_equal(a, b):
                _dump("fused_pyramid_mismatch_%dx%d_l%d" % (shape[1], shape[0], l), fused=a, unfused=b)
            np.testing.assert_array_equal(a, b, err_msg="level %d %s" % (l, what))
    OA = ko.OracleImage(g)
    OA.preprocess(6.0, tiles, tiles)
    for l in range(4):
        oi, od = OA.level(l)
        np.testing.assert_array_equal(planes[1][l][1], oi)
        np.testing.assert_array_equal(planes[1][l][2], od)


def test_small_images_take_the_five_launch_path(mods):
    """A level narrower than 2 * 21 + 2 pixels needs more than one reflection for its border: those sizes keep the per-level
    kernels (their reflect101 loops); the result is still the oracle's."""
    ko, klt = mods
    g = noise_image(320, 240, seed=5)      # level 3 is 40 x 30
    ctx = klt.KltContext(320, 240, 100)
    im = ctx.image(g)
    im.preprocess(6.0, 8, 8)
    OA = ko.OracleImage(g)
    OA.preprocess(6.0, 8, 8)
    _assert_pyramid_equal(im, OA, "small320x240")


def golden_pair_parity(ko, klt, pair, expected, tag):
    """Planes, Harris response, corners, LK positions and status of the HIP path against the committed oracle results
    on one of the golden image pairs; returns (corners detected, points tracked)."""
    a, b = pair
    ctx, HA, HB, OA, OB = _pair(ko, klt, a, b)
    _assert_pyramid_equal(HA, OA, tag + "_a")
    _assert_pyramid_equal(HB, OB, tag + "_b")
    hr = HA.harris()
    orr = ko.harris_response(OA.image)
    if not np.array_equal(hr, orr):
        _dump("harris_mismatch_" + tag, hr=hr, orr=orr)
    np.testing.assert_array_equal(hr, orr)
    kp_h = HA.detect_keypoints(np.zeros((0, 2)), 200, 20.0)
    np.testing.assert_array_equal(kp_h, expected["keypoints"])
    nx_h, st_h = HA.track_keypoints(HB, kp_h, kp_h.copy())
    if not (np.array_equal(st_h, expected["status"]) and np.array_equal(nx_h, expected["next"])):
        _dump("track_mismatch_" + tag, nx_h=nx_h, st_h=st_h, nx_o=expected["next"], st_o=expected["status"])
    np.testing.assert_array_equal(st_h, expected["status"])
    np.testing.assert_array_equal(nx_h, expected["next"])
    return len(kp_h), int(st_h.sum())


def test_golden_pair_v1_full_parity(mods, golden_pair_v1, klt_expected_v1):
    """The reference's two test frames with the first restatement of the undistortion (map rounded straight from the
    double): 165 corners / 162 tracked, one off the reference's known answers.  The pair whose undistortion reproduces
    them exactly is checked in tests/test_zz_golden_pinned_gpu.py."""
    ko, klt = mods
    assert golden_pair_parity(ko, klt, golden_pair_v1, klt_expected_v1, "golden_v1") == (165, 162)


@pytest.mark.parametrize("w,h,n,seed", [(752, 480, 200, 31), (640, 480, 150, 32), (1280, 720, 600, 33)])
def test_detect_and_track_parity_synthetic(mods, w, h, n, seed):
    ko, klt = mods
    g = noise_image(w, h, seed=seed)
    ang = 0.01
    M = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    g2 = warp_affine(g, M, np.array([2.3, -1.7]))
    ctx, HA, HB, OA, OB = _pair(ko, klt, g, g2)
    kp_o = OA.detect_keypoints(np.zeros((0, 2)), n, 20.0)
    kp_h = HA.detect_keypoints(np.zeros((0, 2)), n, 20.0)
    np.testing.assert_array_equal(kp_h, kp_o)
    assert len(kp_o) > n // 2
    for guess in (None, kp_o + np.array([1.0, -0.5])):
        nx_o, st_o = OA.track_keypoints(OB, kp_o, guess)
        nx_h, st_h = HA.track_keypoints(HB, kp_o, guess)
        if not (np.array_equal(st_h, st_o) and np.array_equal(nx_h, nx_o)):
            _dump("track_mismatch_%d" % seed, nx_h=nx_h, st_h=st_h, nx_o=nx_o, st_o=st_o, kp=kp_o)
        np.testing.assert_array_equal(st_h, st_o)
        assert st_o.mean() > 0.8
        np.testing.assert_allclose(nx_h[st_o > 0], nx_o[st_o > 0], rtol=1e-4, atol=0)
        np.testing.assert_array_equal(nx_h, nx_o)
    # second detection with already tracked points present
    kp2_o = OB.detect_keypoints(nx_o[st_o > 0], n, 20.0)
    kp2_h = HB.detect_keypoints(nx_o[st_o > 0], n, 20.0)
    np.testing.assert_array_equal(kp2_h, kp2_o)


def test_detect_paths_prefetch_and_full_list_fallback(mods):
    """The corner pass has three routes to the same answer: launched by detect(), prefetched behind a tracking launch,
    and -- when the greedy spacing pass exhausts the strongest candidates before it has max_points corners -- the
    fallback to the full candidate list.  All must equal the oracle."""
    ko, klt = mods
    w, h = 752, 480
    g = np.full((h, w), 110, np.uint8)
    g[90:390, 200:500] = noise_image(300, 300, seed=77)          # > 1024 NMS survivors packed into 300 x 300 px
    g2 = warp_affine(g, np.eye(2), np.array([1.5, 0.5]))
    # a 150-corner context hands the host's spacing pass the 896 strongest candidates (round 6: 8 per corner beyond 150 corners -- a
    # 600-corner context would hand over all of them and never fall back)
    ctx, HA, HB, OA, OB = _pair(ko, klt, g, g2, max_points=150)
    want = OA.detect_keypoints(np.zeros((0, 2)), 1000, 20.0)      # asks for more corners than 20 px spacing allows
    before = ctx.stats().detect_full_list
    got = HA.detect_keypoints(np.zeros((0, 2)), 1000, 20.0)
    np.testing.assert_array_equal(got, want)
    assert 50 < len(want) < 400
    assert ctx.stats().detect_full_list == before + 1           # the fallback route really ran
    # prefetch: B's Harris pass rides behind the A -> B tracking launch
    kp = want[:100]
    HB.prefetch_detect()
    nx_h, st_h = HA.track_keypoints(HB, kp, None)
    nx_o, st_o = OA.track_keypoints(OB, kp, None)
    np.testing.assert_array_equal(st_h, st_o)
    np.testing.assert_array_equal(nx_h, nx_o)
    np.testing.assert_array_equal(HB.detect_keypoints(nx_o[st_o > 0], 150, 20.0), OB.detect_keypoints(nx_o[st_o > 0], 150, 20.0))
    # and again without a hint on the same image: same result
    np.testing.assert_array_equal(HB.detect_keypoints(nx_o[st_o > 0], 150, 20.0), OB.detect_keypoints(nx_o[st_o > 0], 150, 20.0))


def test_candidate_selection_edge_cases_few_candidates_and_massive_ties(mods):
    """k_harris_select orders the strongest candidates on the device (response bins, rank inside a bin: ties broken by the pixel index).
    (a) Fewer NMS survivors than it is asked to keep: all of them are handed over, in order.  (b) A periodic image (period 4, CLAHE tiles
    that are multiples of it: identical lookup tables) gives thousands of candidates with IDENTICAL responses -- one bin, more entries than
    the top block holds: the host fetches the full list, and the visiting order is decided by the index alone.  Both equal the oracle."""
    ko, klt = mods
    w, h = 752, 480
    # (a) a flat frame with one small textured patch
    g = np.full((h, w), 110, np.uint8)
    g[200:240, 300:340] = noise_image(40, 40, seed=5)
    ctx, HA, HB, OA, OB = _pair(ko, klt, g, g, max_points=150)
    want = OA.detect_keypoints(np.zeros((0, 2)), 150, 20.0)
    before = ctx.stats().detect_full_list
    np.testing.assert_array_equal(HA.detect_keypoints(np.zeros((0, 2)), 150, 20.0), want)
    assert 0 < len(want) < 20
    assert ctx.stats().detect_full_list == before          # everything fitted the top block
    # (b) period-4 texture, 4 x 4 CLAHE tiles of 188 x 120 pixels
    cell = np.random.RandomState(9).randint(60, 200, (4, 4)).astype(np.uint8)
    p = np.tile(cell, (h // 4, w // 4))
    ctx, HA, HB, OA, OB = _pair(ko, klt, p, p, tiles=4, max_points=150)
    resp = ko.harris_response(OA.image)
    vals, counts = np.unique(resp[40:-40, 40:-40], return_counts=True)
    assert counts.max() > 8192                                # thousands of exactly equal responses away from the border
    want = OA.detect_keypoints(np.zeros((0, 2)), 150, 20.0)
    before = ctx.stats().detect_full_list
    got = HA.detect_keypoints(np.zeros((0, 2)), 150, 20.0)
    np.testing.assert_array_equal(got, want)
    assert len(want) > 50
    # with a few existing points in the way as well
    have = want[::7] + np.array([3.0, -2.0])
    np.testing.assert_array_equal(HA.detect_keypoints(have, 150, 20.0), OA.detect_keypoints(have, 150, 20.0))
    assert ctx.stats().detect_full_list >= before             # (whether the fallback ran depends on where the boundary bin falls)


def test_plain_lk_parity_including_failures(mods):
    ko, klt = mods
    g = noise_image(752, 480, seed=41)
    g2 = warp_affine(g, np.eye(2), np.array([-4.0, 3.0]))
    g2[:, 600:] = 90        # textureless strip -> minEig rejections
    g[:, 600:] = 90
    ctx, HA, HB, OA, OB = _pair(ko, klt, g, g2)
    rng = np.random.RandomState(5)
    pts = (rng.rand(300, 2) * [800, 520] - [24, 20]).astype(np.float32)   # includes out-of-image points
    guess = pts + rng.randn(300, 2).astype(np.float32) * 3
    nx_o, st_o, _ = OA.lk(OB, pts, guess)
    nx_h, st_h = HA.lk(HB, pts, guess)
    if not (np.array_equal(st_h, st_o) and np.array_equal(nx_h, nx_o)):
        _dump("lk_plain_mismatch", nx_h=nx_h, st_h=st_h, nx_o=nx_o, st_o=st_o, pts=pts, guess=guess)
    np.testing.assert_array_equal(st_h, st_o)
    assert 0 < st_o.sum() < len(st_o)
    np.testing.assert_array_equal(nx_h, nx_o)


def test_track_edge_cases(mods):
    ko, klt = mods
    g = noise_image(640, 480, seed=51)
    ctx, HA, HB, OA, OB = _pair(ko, klt, g, g.copy())
    nx, st = HA.track_keypoints(HB, np.zeros((0, 2)), None)          # empty input
    assert len(st) == 0
    pts = np.array([[5.0, 5.0], [635.0, 475.0], [19.9, 240.0], [320.0, 240.0], [-30.0, 100.0], [700.0, 100.0],
                    [20.0, 20.0], [619.99, 459.99]])
    nx_o, st_o = OA.track_keypoints(OB, pts, pts.copy())
    nx_h, st_h = HA.track_keypoints(HB, pts, pts.copy())
    np.testing.assert_array_equal(st_h, st_o)
    np.testing.assert_array_equal(nx_h, nx_o)
    flat = np.full((480, 640), 100, np.uint8)
    ctx2, FA, FB, _, _ = _pair(ko, klt, flat, flat)
    _, stf = FA.track_keypoints(FB, pts, pts.copy())
    assert not stf.any()
    assert len(FA.detect_keypoints(np.zeros((0, 2)), 100, 20.0)) == 0   # no corners on a flat image


def test_error_behaviour(mods):
    ko, klt = mods
    from xrslam_amd._lib import XrhipError, XRHIP_ESTATE, XRHIP_EINVAL
    ctx = klt.KltContext(640, 480, 100)
    im = ctx.image()
    with pytest.raises(XrhipError) as e:
        im.preprocess()
    assert e.value.code == XRHIP_ESTATE
    im.upload(noise_image(640, 480, seed=1))
    with pytest.raises(XrhipError) as e:
        im.detect_keypoints(np.zeros((0, 2)), 10, 20.0)
    assert e.value.code == XRHIP_ESTATE
    with pytest.raises(XrhipError) as e:
        klt.KltContext(10, 10, 10)
    assert e.value.code == XRHIP_EINVAL
    im.preprocess()
    im.release_image_buffer()
    with pytest.raises(XrhipError):
        im.detect_keypoints(np.zeros((0, 2)), 10, 20.0)


def test_full_size_properties(mods):
    """Size-independent properties at the BASELINE workload size (752x480, 150 pts):
    tracking an image onto itself is the identity; a pure translation is recovered."""
    ko, klt = mods
    g = noise_image(752, 480, seed=61)
    shift = np.array([5.5, 3.25])
    g2 = warp_affine(g, np.eye(2), -shift)
    ctx = klt.KltContext(752, 480, 150)
    A, B, C2 = ctx.image(g), ctx.image(g), ctx.image(g2)
    for im in (A, B, C2):
        im.preprocess()
    kp = A.detect_keypoints(np.zeros((0, 2)), 150, 20.0)
    assert 100 < len(kp) <= 150
    nx, st = A.track_keypoints(B, kp, kp.copy())
    assert st.all() and np.abs(nx - kp).max() < 1e-3
    nx, st = A.track_keypoints(C2, kp, None)
    ok = st > 0
    assert ok.mean() > 0.9
    assert np.median(np.abs(nx[ok] - kp[ok] - shift)) < 0.05


def test_device_undistortion_bit_exact(mods):
    """cv::undistort / ImageUndistorter on the device (k_undistort) against oracle/undistort.py -- the
    EuRoC radial-tangential lens at 752x480 and the TUM-VI equidistant fisheye at 512x512, host and HBM-resident input;
    the map comes from the oracle's own arithmetic, not from the product's builder."""
    import ctypes as C
    from oracle import undistort as ou
    _, klt = mods
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    cases = [("cv_undistort", 752, 480, (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)),
             ("equidistant", 512, 512, (190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504),
              (0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182))]
    for model, w, h, K, D in cases:
        img = noise_image(w, h, seed=31 + w)
        exp = ou.undistort(img, K, D) if model == "cv_undistort" else ou.undistort_model(img, K, D, model)
        ctx = klt.KltContext(w, h, 150)
        ctx.set_undistort_map(ou.packed_map(w, h, K, D, model))
        im = ctx.image()
        im.upload_distorted(img)
        np.testing.assert_array_equal(im.raw(), exp)
        # the same frame already resident in HBM (allocated through the HIP runtime the library itself is linked to: a
        # second runtime in the process -- torch brings its own copy -- cannot be initialised after the first)
        dev = C.c_void_p()
        assert hip.hipMalloc(C.byref(dev), C.c_size_t(img.size)) == 0
        assert hip.hipMemcpy(dev, C.c_void_p(img.ctypes.data), C.c_size_t(img.size), 1) == 0   # hipMemcpyHostToDevice
        im2 = ctx.image()
        im2.upload_distorted_device(dev.value, w)
        np.testing.assert_array_equal(im2.raw(), exp)
        ctx.synchronize()
        assert hip.hipFree(dev) == 0
        assert (exp != img).mean() > 0.5                     # the lens model does move the pixels
        # and the rectified frame feeds the same preprocessing as a host-rectified one
        im.preprocess()
        ref = ctx.image(exp)
        ref.preprocess()
        np.testing.assert_array_equal(im.level(0)[0], ref.level(0)[0])
        ctx.set_undistort_map(None)
        with pytest.raises(Exception):
            im.upload_distorted(img)
