"""Pre-integration records of a fixed set of intervals (1 .. 161 samples) as a .npy -- run once per library (XRSLAM_HIP_LIB) on a GPU box and
compare: round 6's split chains against the round-5 library (xrslam_amd/lib/libxrslam_hip_r5.so) gave bit-identical deltas and bias
Jacobians on all 13 cases (the covariance factor differs by 1.3e-16 relative: the new Cholesky block).
    python tools/preint_bits.py gpurun_out/preint_new.npy; XRSLAM_HIP_LIB=... python tools/preint_bits.py gpurun_out/preint_r5.npy"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from tests import ba_synth as bs
from xrslam_amd import ba
out = sys.argv[1]
ctx = ba.BaContext()
res = []
pd, truth = bs.make_window(K=5, L=20, seed=31)
for k, smp in enumerate(truth["samples"]):
    res.append(ctx.preintegrate(smp, truth["times"][k + 1], pd.frame_state[k, 10:13], pd.frame_state[k, 13:16], bs.NOISE36, True, True))
for n in (1, 11, 31, 32, 33, 64, 65, 100, 161):
    rng = np.random.RandomState(100 + n)
    smp = np.zeros((n, 7)); smp[:, 0] = 2.0 + 0.005 * np.arange(n)
    smp[:, 1:4] = 0.3 * rng.randn(n, 3) + np.array([0.2, -0.1, 0.4]); smp[:, 4:7] = np.array([0.3, -0.2, 9.7]) + 0.5 * rng.randn(n, 3)
    bg, b2 = 1e-3 * rng.randn(3), 1e-2 * rng.randn(3)
    res.append(ctx.preintegrate(smp, float(smp[-1, 0] + 0.003), bg, b2, bs.NOISE36, True, n > 1))
np.save(out, np.array(res))
