"""Regenerates the committed golden fixtures from the reference's own test data.

Runs ONLY in the build container (needs /root/reference); the GPU box uses the
committed .npz files.  Inputs: the two EuRoC V1_01 frames used by
xrslam-test/test/src/test_feature_track.cpp:27-28, undistorted exactly as that
test does (cv::undistort with the intrinsics at :10-22, restated in
oracle/undistort.py).  Outputs (tests/golden/):
  euroc_pair.npz        undistorted frames (uint8 480x752) a, b -- the restatement that reproduces the reference's known
                        answers exactly (map positions rounded through float32, see oracle/undistort.py)
  klt_expected.npz      oracle results on that pair: detected keypoints, tracked positions / status -- regression pins
                        for the oracle and the parity target of the HIP path
  euroc_pair_v1.npz,    the same with the map positions rounded straight from the double (the round-1 fixtures before
  klt_expected_v1.npz   the pin was found; 36 pixels differ): a second image pair for the HIP-vs-oracle parity test
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import klt_oracle as ko  # noqa: E402
from oracle.undistort import undistort  # noqa: E402

REF = "/root/reference/xrslam-test/data/"
K = (458.654, 457.296, 367.215, 248.375)
D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def one(precision, suffix):
    a = undistort(np.array(Image.open(REF + "1403715282262142976.png")), K, D, precision)
    b = undistort(np.array(Image.open(REF + "1403715282312143104.png")), K, D, precision)
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, "euroc_pair%s.npz" % suffix), a=a, b=b)
    A = ko.OracleImage(a)
    B = ko.OracleImage(b)
    A.preprocess(6.0, 8, 8)
    B.preprocess(6.0, 8, 8)
    kp = A.detect_keypoints(np.zeros((0, 2)), 200, 20.0)
    nx, st = A.track_keypoints(B, kp, kp.copy())   # identity prediction
    np.savez_compressed(os.path.join(here, "klt_expected%s.npz" % suffix), keypoints=kp, next=nx, status=st,
                        clahe_a_crc=np.array([int(A.image.astype(np.uint64).sum())]))
    print(precision, "detected", len(kp), "tracked", int(st.sum()))


def main():
    one("float32", "")
    one("float64", "_v1")


if __name__ == "__main__":
    main()
