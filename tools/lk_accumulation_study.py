"""Quantifies deviation #1 of DESIGN.md section 5: the LK window sums (A11, A12, A22, b1, b2) are exact integers here,
while OpenCV's calcOpticalFlowPyrLK accumulates them in float -- one pixel after the other in its scalar path, in four
strided lanes in its 128-bit SIMD path -- so its result depends on the build.  The study runs the CPU oracle's
track_keypoints (forward + backward LK, gates, 0.5 px round trip: opencv_image.cpp:75-154) with the three accumulations on
the reference's own two EuRoC frames and on synthetic pairs, and reports how many status bits flip and how far positions move.

    python tools/lk_accumulation_study.py          # needs oracle/_build (make -C oracle); no GPU
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import klt_oracle as ko  # noqa: E402  (development aid; never imported by the product)
from tests.util import noise_image, warp_affine  # noqa: E402


def run(a, b, n_points, guess_noise=0.0, seed=0):
    A, B = ko.OracleImage(a), ko.OracleImage(b)
    A.preprocess()
    B.preprocess()
    kp = A.detect_keypoints(np.zeros((0, 2)), n_points, 20.0)
    rng = np.random.RandomState(seed)
    guess = None if guess_noise == 0 else kp + rng.randn(*kp.shape) * guess_noise
    out = {}
    for mode in (0, 1, 2):
        ko.lib().orc_set_lk_accumulation(mode)
        out[mode] = A.track_keypoints(B, kp, guess)
    ko.lib().orc_set_lk_accumulation(0)
    return kp, out


def report(name, kp, out):
    p0, s0 = out[0]
    line = "%-34s %4d points, %4d tracked |" % (name, len(kp), int(s0.sum()))
    for mode, label in ((1, "float raster"), (2, "float 4-lane")):
        p, s = out[mode]
        both = (s0 == 1) & (s == 1)
        flips = int((s0 != s).sum())
        dmax = float(np.abs(p[both] - p0[both]).max()) if both.any() else 0.0
        line += " %s: %d status flips, max |dp| %.2e px |" % (label, flips, dmax)
    print(line)
    return line


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "euroc_pair.npz"))
    lines = [report("EuRoC test frames (reference fixture)", *run(z["a"], z["b"], 200))]
    for k, (shift, rot) in enumerate([((1.5, -2.25), 0.0), ((6.0, 3.0), 0.02), ((14.0, -9.0), -0.03)]):
        g = noise_image(752, 480, seed=70 + k)
        c, s = np.cos(rot), np.sin(rot)
        g2 = warp_affine(g, np.array([[c, -s], [s, c]]), np.array(shift))
        lines.append(report("synthetic 752x480, shift %s rot %.2f" % (shift, rot), *run(g, g2, 300)))
        lines.append(report("  same, guesses within 2 px", *run(g, g2, 300, guess_noise=2.0, seed=k)))
    return lines


if __name__ == "__main__":
    main()
