"""Development aid: wall-clock per xrhip_ba_preintegrate call for a range of segment lengths (needs an MI355X)."""
import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xrslam_amd import ba
ctx = ba.BaContext()
rng = np.random.RandomState(0)
noise = np.zeros(36); 
for b in range(4): noise[9*b+0]=noise[9*b+4]=noise[9*b+8]=[1e-4,1e-2,1e-6,1e-5][b]
for n in (2, 10, 20, 40, 80):
    smp = np.zeros((n,7)); smp[:,0] = 0.005*np.arange(n); smp[:,1:4] = 0.1*rng.randn(n,3); smp[:,4:7] = [0,0,9.8]+0.1*rng.randn(n,3)
    for jac,cov in ((0,0),(1,1)):
        for _ in range(20): ctx.preintegrate(smp, 0.005*n, np.zeros(3), np.zeros(3), noise, jac, cov)
        t0=time.perf_counter()
        for _ in range(200): ctx.preintegrate(smp, 0.005*n, np.zeros(3), np.zeros(3), noise, jac, cov)
        print("n=%3d jac/cov=%d: %.1f us per call"%(n, cov, (time.perf_counter()-t0)/200*1e6))
