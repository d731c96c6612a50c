"""CPU tests of the headless player's I/O (xrslam_amd/csrc/player/euroc_io.hpp):
PNG decode, radial-tangential undistortion, the TUM-VI reader's equidistant undistortion, ASL/EuRoC CSV parsing,
event order, TUM / CSV output, ATE, and the command line the reference's player accepts."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_check", "player_host.cpp")
OUT = os.path.join(ROOT, "tests", "host_check", "_build", "libplayer_host.so")


@pytest.fixture(scope="module")
def ph():
    deps = [SRC, os.path.join(ROOT, "xrslam_amd", "csrc", "player", "euroc_io.hpp"),
            os.path.join(ROOT, "xrslam_amd", "csrc", "host", "hla.hpp")]
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", OUT, "-lz"])
    lib = C.CDLL(OUT)
    lib.ph_ate.restype = C.c_double
    lib.ph_merge.restype = C.c_long
    lib.ph_load_truth.restype = C.c_long
    lib.ph_merge.argtypes = [C.c_char_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    lib.ph_load_truth.argtypes = [C.c_char_p, C.c_void_p, C.c_long]
    lib.ph_ate.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    lib.ph_tum_line.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_char_p, C.c_long]
    lib.ph_decode_png.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    lib.ph_undistort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ph_undistort_model.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
    lib.ph_csv_line.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_char_p, C.c_long]
    lib.ph_split_url.argtypes = [C.c_char_p, C.c_char_p, C.c_long]
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _decode(ph, path):
    w, h = C.c_int(), C.c_int()
    assert ph.ph_decode_png(path.encode(), C.byref(w), C.byref(h), None, 0) == 0
    out = np.zeros((h.value, w.value), np.uint8)
    assert ph.ph_decode_png(path.encode(), C.byref(w), C.byref(h), _p(out), out.size) == 0
    return out


def test_png_decoder_matches_pil(ph, tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    smooth = (np.add.outer(np.arange(97), 2 * np.arange(131)) % 256).astype(np.uint8)   # exercises the predictive filters
    noise = rng.randint(0, 256, (60, 75)).astype(np.uint8)
    for k, (arr, kw) in enumerate([(smooth, {}), (noise, {}), (smooth, {"optimize": True}), (noise, {"compress_level": 1})]):
        path = str(tmp_path / ("g%d.png" % k))
        Image.fromarray(arr).save(path, **kw)
        np.testing.assert_array_equal(_decode(ph, path), arr)
    # 16-bit grey (TUM-VI): the high byte, like cv::imread(IMREAD_GRAYSCALE) / png_set_strip_16
    deep = (rng.randint(0, 65536, (33, 47)).astype(np.uint16) & 0xFFF0) + (np.add.outer(np.arange(33), np.arange(47)) % 16).astype(np.uint16)
    path = str(tmp_path / "g16.png")
    Image.fromarray(deep).save(path)
    assert Image.open(path).mode in ("I;16", "I;16B", "I")
    np.testing.assert_array_equal(_decode(ph, path), (deep >> 8).astype(np.uint8))
    rgb = rng.randint(0, 256, (40, 50, 3)).astype(np.uint8)
    path = str(tmp_path / "rgb.png")
    Image.fromarray(rgb).save(path)
    r64 = rgb.astype(np.int64)
    want = ((r64[..., 0] * 4899 + r64[..., 1] * 9617 + r64[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    np.testing.assert_array_equal(_decode(ph, path), want)
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png")
    w, h = C.c_int(), C.c_int()
    assert ph.ph_decode_png(str(bad).encode(), C.byref(w), C.byref(h), None, 0) != 0


def test_undistort_matches_the_oracle_restatement(ph):
    from oracle import undistort as ou
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (120, 188)).astype(np.uint8)
    K = np.array([458.654 / 4, 457.296 / 4, 367.215 / 4, 248.375 / 4])
    D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    out = np.zeros_like(img)
    ph.ph_undistort(_p(img), img.shape[1], img.shape[0], _p(K), _p(D), _p(out))
    np.testing.assert_array_equal(out, ou.undistort(img, K, D))
    out0 = np.zeros_like(img)
    ph.ph_undistort(_p(img), img.shape[1], img.shape[0], _p(K), _p(np.zeros(4)), _p(out0))
    np.testing.assert_array_equal(out0, img)          # zero distortion is the identity map


def _tiny_dir(tmp_path):
    root = tmp_path / "mav0"
    (root / "cam0").mkdir(parents=True)
    (root / "imu0").mkdir()
    (root / "state_groundtruth_estimate0").mkdir()
    cam_t = [1403636579763555584, 1403636579813555456, 1403636579863555584]
    (root / "cam0" / "data.csv").write_text("#timestamp [ns],filename\r\n" +
                                            "".join("%d,%d.png\r\n" % (t, t) for t in cam_t))
    imu_t = [cam_t[0] - 5000000 + 5000000 * k for k in range(24)]
    (root / "imu0" / "data.csv").write_text("#timestamp [ns],w,w,w,a,a,a\r\n" +
                                            "".join("%d,%g,%g,%g,%g,%g,%g\r\n" % (t, .1, .2, .3, 9.8, 0., k) for k, t in enumerate(imu_t)))
    (root / "state_groundtruth_estimate0" / "data.csv").write_text(
        "#timestamp, p, q, v, bw, ba\r\n" + "".join(
            "%d,%g,%g,%g,1,0,0,0,%g,0,0,0.001,0,0,0.01,0,0\r\n" % (t, 0.1 * k, 0.0, 1.0, 0.5) for k, t in enumerate(cam_t)))
    return str(root), cam_t, imu_t


def test_event_order_and_csv_parsing(ph, tmp_path):
    root, cam_t, imu_t = _tiny_dir(tmp_path)
    types = np.zeros(256, np.int32); index = np.zeros(256, np.int64); times = np.zeros(256)
    n = ph.ph_merge(root.encode(), 0.0, _p(types), _p(index), _p(times), 256)
    assert n == len(cam_t) + 2 * len(imu_t)
    types, index, times = types[:n], index[:n], times[:n]
    assert np.all(np.diff(times) >= 0)
    # at an equal time stamp: gyroscope, accelerometer, then the camera (async_dataset_reader.cpp:16-49)
    k = imu_t.index(cam_t[0])
    at = np.nonzero(np.abs(times - cam_t[0] * 1e-9) < 1e-12)[0]
    assert list(types[at]) == [0, 1, 2] and list(index[at]) == [k, k, 0]
    # a positive camera time offset moves the frame behind the samples that now precede it
    n2 = ph.ph_merge(root.encode(), 0.004, _p(np.zeros(256, np.int32)), _p(np.zeros(256, np.int64)), _p(times), 256)
    assert n2 == n
    rows = np.zeros((8, 17))
    m = ph.ph_load_truth((root + "/state_groundtruth_estimate0/data.csv").encode(), _p(rows), 8)
    assert m == 3
    np.testing.assert_allclose(rows[1, :8], [cam_t[1] * 1e-9, 0.1, 0.0, 1.0, 1, 0, 0, 0])
    np.testing.assert_allclose(rows[1, 8:], [0.5, 0, 0, 0.001, 0, 0, 0.01, 0, 0])


def test_tum_line_and_ate(ph):
    buf = C.create_string_buffer(512)
    p = np.array([1.5, -2.25, 3.0]); q = np.array([0.0, 0.0, 0.6, 0.8])
    ph.ph_tum_line(1403636579.763555584, _p(p), _p(q), buf, 512)
    want = "%.18e %.9e %.9e %.9e %.7e %.7e %.7e %.7e\n" % ((1403636579.763555584,) + tuple(p) + tuple(q))
    assert buf.value.decode() == want                    # trajectory_writer.h:54-76
    from xrslam_amd.harness import runner
    rng = np.random.RandomState(3)
    ref = rng.randn(50, 3)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    est = (ref - [1, 2, 3]) @ R + 0.01 * rng.randn(50, 3)
    got = ph.ph_ate(_p(np.ascontiguousarray(est)), _p(np.ascontiguousarray(ref)), 50)
    poses = [(float(i), *est[i], 0, 0, 0, 1) for i in range(50)]
    seq = {"cam_t": np.arange(50.0), "states": np.concatenate([np.tile([0, 0, 0, 1.0], (50, 1)), ref, np.zeros((50, 9))], 1)}
    try:
        want_ate = runner.ate_rmse(poses, seq)
    except Exception:
        want_ate = None
    assert 0.005 < got < 0.03
    if want_ate is not None and np.isfinite(want_ate):
        assert abs(got - want_ate) < 1e-9


def test_model_undistorter_matches_the_oracle_restatement(ph):
    """xrslam::extra::ImageUndistorter (image_undistorter.h:14-92), the TUM-VI reader's undistortion
    (tum_dataset_reader.cpp:67-76): equidistant and radtan maps against the numpy restatement."""
    from oracle import undistort as ou
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (128, 128)).astype(np.uint8)
    # TUM-VI cam0 (512x512: fx 190.98, fy 190.97, cx 254.93, cy 256.90; k1..k4) scaled to 128x128
    K = np.array([190.978 / 4, 190.973 / 4, 254.932 / 4, 256.897 / 4])
    D = np.array([0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182])
    out = np.zeros_like(img)
    assert ph.ph_undistort_model(_p(img), 128, 128, _p(K), _p(D), 4, b"equidistant", _p(out)) == 0
    np.testing.assert_array_equal(out, ou.undistort_model(img, K, D, "equidistant"))
    assert np.count_nonzero(out != img) > img.size // 2          # a fisheye map is far from the identity
    # the principal point maps to itself (r < 1e-10 branch): put it on a pixel centre
    Kc = np.array([48.0, 48.0, 64.0, 64.0])
    assert ph.ph_undistort_model(_p(img), 128, 128, _p(Kc), _p(D), 4, b"equidistant", _p(out)) == 0
    np.testing.assert_array_equal(out, ou.undistort_model(img, Kc, D, "equidistant"))
    assert out[64, 64] == img[64, 64]
    # radtan with the optional k3
    D5 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.01])
    K2 = np.array([458.654 / 6, 457.296 / 6, 367.215 / 6, 248.375 / 6])
    img2 = rng.randint(0, 256, (80, 125)).astype(np.uint8)
    out2 = np.zeros_like(img2)
    for d in (D5, D5[:4]):
        assert ph.ph_undistort_model(_p(img2), 125, 80, _p(K2), _p(np.ascontiguousarray(d)), len(d), b"radtan", _p(out2)) == 0
        np.testing.assert_array_equal(out2, ou.undistort_model(img2, K2, d, "radtan"))
    # zero coefficients: both models are the identity map
    for model in (b"radtan", b"equidistant"):
        assert ph.ph_undistort_model(_p(img), 128, 128, _p(K), _p(np.zeros(4)), 4, model, _p(out)) == 0
        if model == b"radtan":
            np.testing.assert_array_equal(out, img)
    assert ph.ph_undistort_model(_p(img), 128, 128, _p(K), _p(D), 4, b"fov", _p(out)) == 1    # "unknown model" throws


def test_csv_line_and_dataset_urls(ph):
    buf = C.create_string_buffer(512)
    p = np.array([1.5, -2.25, 3.0]); q = np.array([0.0, 0.0, 0.6, 0.8])
    ph.ph_csv_line(1403636579.763555584, _p(p), _p(q), buf, 512)
    want = "%.18e,%.9e,%.9e,%.9e,%.7e,%.7e,%.7e,%.7e\n" % ((1403636579.763555584,) + tuple(p) + tuple(q))
    assert buf.value.decode() == want                    # trajectory_writer.h:46-52
    path = C.create_string_buffer(256)
    assert ph.ph_split_url(b"euroc:///data/MH_01_easy/mav0", path, 256) == 1 and path.value == b"/data/MH_01_easy/mav0"
    assert ph.ph_split_url(b"tum://room1/mav0", path, 256) == 2 and path.value == b"room1/mav0"
    assert ph.ph_split_url(b"/data/MH_01_easy/mav0", path, 256) == 0          # dataset_reader.cpp:17-34: no reader
    assert ph.ph_split_url(b"eur", path, 256) == 0


PLAYER = os.path.join(ROOT, "xrslam_amd", "bin", "xrslam-player")


@pytest.mark.skipif(not os.path.exists(PLAYER), reason="player not built (run __graft_entry__.build())")
def test_player_accepts_the_reference_command_line(tmp_path):
    """xrslam-pc/player/src/main.cpp:57-79: -sc / -dc / -lc / --tum / --csv / -p and a positional euroc:// or tum://
    input.  No device is needed up to the point where the data set is opened."""
    slam, dev = os.path.join(ROOT, "configs", "euroc_slam.yaml"), os.path.join(ROOT, "configs", "euroc_sensor.yaml")
    run = lambda *a: subprocess.run([PLAYER, *a], capture_output=True, text=True)
    r = run()
    assert r.returncode == 2 and "usage" in r.stderr
    r = run("-sc", slam, "-dc", dev, "-p", str(tmp_path))                 # no scheme: the reference finds no reader
    assert r.returncode == 1 and 'Cannot open "%s"' % tmp_path in r.stderr
    for url in ("euroc://" + str(tmp_path), "tum://" + str(tmp_path)):
        r = run("--slamconfig", slam, "--deviceconfig", dev, "--license", "none", "--tum", str(tmp_path / "t.tum"),
                "--csv", str(tmp_path / "t.csv"), "--play", url)
        assert r.returncode == 1 and "no camera or IMU data under %s" % tmp_path in r.stderr
    r = run("-sc", slam, "-dc", dev, "-q", "euroc://x")
    assert r.returncode == 2 and "unknown argument -q" in r.stderr
