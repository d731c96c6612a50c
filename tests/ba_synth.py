"""Seeded synthetic sliding-window VI-BA problems (numpy), shared by CPU and GPU tests and bench.py.

Scene: camera-IMU rig (EuRoC extrinsics/intrinsics, configs/euroc_sensor.yaml:14-53) moving on a
smooth figure-eight; random landmarks 2-8 m ahead; 200 Hz IMU from the analytic trajectory;
keyframes every `kf_dt` seconds.  Produces xrslam_amd.abi.BaProblemData in exactly the layout the
reference's refine_window / localize_newframe assemble.
"""
import numpy as np

from xrslam_amd import abi

GRAVITY = 9.80665
K_EUROC = (458.654, 457.296, 367.215, 248.375)
Q_BC = np.array([-7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01])
P_BC = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])
CAM_EXT = np.concatenate([Q_BC, P_BC])
IMU_EXT = np.array([0, 0, 0, 1.0, 0, 0, 0])
SQRT_INV_COV = np.array([K_EUROC[0], K_EUROC[1]]) / np.sqrt(0.5)
NOISE36 = np.concatenate([np.eye(3).ravel() * 2.8791302399999997e-08, np.eye(3).ravel() * 4.0e-6,
                          np.eye(3).ravel() * 3.7608844899999997e-10, np.eye(3).ravel() * 9.0e-6])


from xrslam_amd.harness.trajectory import Trajectory, qconj, qexp, qlog, qmat, qmul, qrot  # noqa: E402,F401


def make_window(K=10, L=150, seed=1, kf_dt=0.25, imu_hz=200.0, pixel_noise=0.5, state_noise=1.0, with_prior=True,
                max_iterations=30, n_fixed_first=0):
    """Returns (problem: abi.BaProblemData at perturbed states, truth dict)."""
    from oracle import ba_oracle as bo   # checker-side pre-integration; tests/bench only
    rng = np.random.RandomState(seed)
    traj = Trajectory()
    bg_true = rng.randn(3) * 1e-3
    ba_true = rng.randn(3) * 1e-2
    times = 5.0 + kf_dt * np.arange(K)
    # truth states
    st = np.zeros((K, 16))
    for k, t in enumerate(times):
        st[k, 0:4] = traj.q(t)
        st[k, 4:7] = traj.p(t)
        st[k, 7:10] = traj.v(t)
        st[k, 10:13] = bg_true
        st[k, 13:16] = ba_true
    # camera poses
    def cam_pose(s):
        return qmul(s[0:4], Q_BC), s[4:7] + qrot(s[0:4], P_BC)

    fx, fy, cx, cy = K_EUROC
    # landmarks: sample pixels in random keyframes, back-project to random depth
    pts = []
    while len(pts) < L:
        k = rng.randint(K)
        qc, pc = cam_pose(st[k])
        u = np.array([rng.uniform(30, 722), rng.uniform(30, 450)])
        ray = np.array([(u[0] - cx) / fx, (u[1] - cy) / fy, 1.0])
        ray /= np.linalg.norm(ray)
        pts.append(pc + qrot(qc, ray) * rng.uniform(2.0, 8.0))
    pts = np.array(pts)
    obs = dict(tgt=[], ref=[], lm=[], z_tgt=[], z_ref=[])
    inv_depth = np.zeros(L)
    for l in range(L):
        seen = []
        for k in range(K):
            qc, pc = cam_pose(st[k])
            y = qrot(qconj(qc), pts[l] - pc)
            if y[2] < 0.2:
                continue
            px = np.array([y[0] / y[2] * fx + cx, y[1] / y[2] * fy + cy])
            if px[0] < 20 or px[0] > 732 or px[1] < 20 or px[1] > 460:
                continue
            px = px + rng.randn(2) * pixel_noise
            z = np.array([(px[0] - cx) / fx, (px[1] - cy) / fy, 1.0])
            seen.append((k, z / np.linalg.norm(z), np.linalg.norm(y)))
        if len(seen) < 2:
            inv_depth[l] = 0.25
            continue
        ref_k, z_ref, dist = seen[0]
        inv_depth[l] = 1.0 / dist
        for (k, z, _) in seen[1:]:
            obs["tgt"].append(k)
            obs["ref"].append(ref_k)
            obs["lm"].append(l)
            obs["z_tgt"].append(z)
            obs["z_ref"].append(z_ref)
    # IMU factors (j-1 -> j), pre-integrated at the (noisy) bias estimates
    bg_est = bg_true + rng.randn(3) * 2e-5 * state_noise
    ba_est = ba_true + rng.randn(3) * 3e-4 * state_noise
    imu = dict(i=[], j=[], data=[])
    samples_all = []
    for k in range(1, K):
        t0, t1 = times[k - 1], times[k]
        ts = t0 + np.arange(int(round((t1 - t0) * imu_hz))) / imu_hz
        smp = np.zeros((len(ts), 7))
        for n, t in enumerate(ts):
            w, a = traj.imu(t + 0.5 / imu_hz, bg_true, ba_true)
            smp[n] = np.concatenate([[t], w + rng.randn(3) * 1.7e-4 * np.sqrt(imu_hz) * 0.0,
                                     a + rng.randn(3) * 2e-3 * np.sqrt(imu_hz) * 0.0])
        samples_all.append(smp)
        imu["i"].append(k - 1)
        imu["j"].append(k)
        imu["data"].append(bo.preintegrate(smp, t1, bg_est, ba_est, NOISE36))
    # perturbed start
    x0 = st.copy()
    for k in range(K):
        if k < n_fixed_first:
            continue
        x0[k, 0:4] = qmul(st[k, 0:4], qexp(rng.randn(3) * 0.002 * state_noise))
        x0[k, 4:7] += rng.randn(3) * 0.004 * state_noise
        x0[k, 7:10] += rng.randn(3) * 0.01 * state_noise
        x0[k, 10:13] = bg_est
        x0[k, 13:16] = ba_est
    d0 = inv_depth * (1 + rng.randn(L) * 0.03 * state_noise)
    fix = np.zeros(K, np.uint8)
    fix[:n_fixed_first] = abi.FIX_POSE | abi.FIX_MOTION
    prior = None
    if with_prior:
        n = 15 * (K - 1)
        si = np.zeros((n, n))
        si[0:3, 0:3] = 1.0e15 * np.eye(3)     # marginalization_factor.h:31-32 (gauge prior on the first pose)
        si[3:6, 3:6] = 1.0e15 * np.eye(3)
        prior = dict(frames=np.arange(K - 1), sqrt_info=si, infovec=np.zeros(n), lin=x0[:K - 1].copy())
    pd = abi.BaProblemData(x0, fix, CAM_EXT, IMU_EXT, SQRT_INV_COV, d0, None, obs=obs, imu=imu, prior=prior,
                           max_iterations=max_iterations)
    truth = dict(states=st, inv_depth=inv_depth, points=pts, times=times, samples=samples_all, bg=bg_true, ba=ba_true)
    return pd, truth


def make_localize(seed=2, L=120, pixel_noise=0.5):
    """localize_newframe-shaped problem (sliding_window_tracker.cpp:119-143): frame 0 = previous frame
    (constant), frame 1 = new frame (free), reprojection PRIOR factors (ref pose + depth constant)."""
    pd, truth = make_window(K=4, L=L, seed=seed, with_prior=False, pixel_noise=pixel_noise)
    K = len(pd.frame_state)
    j = K - 1
    fix = np.full(K, abi.FIX_POSE | abi.FIX_MOTION, np.uint8)
    fix[j] = 0
    keep = pd.obs_tgt == j
    obs = dict(tgt=pd.obs_tgt[keep], ref=pd.obs_ref[keep], lm=pd.obs_lm[keep], z_tgt=pd.obs_z_tgt[keep],
               z_ref=pd.obs_z_ref[keep])
    states = truth["states"].copy()
    states[j] = pd.frame_state[j]
    ki = np.where(pd.imu_j == j)[0]
    imu = dict(i=pd.imu_i[ki], j=pd.imu_j[ki], data=pd.imu_data[ki])
    out = abi.BaProblemData(states, fix, CAM_EXT, IMU_EXT, SQRT_INV_COV, truth["inv_depth"],
                            np.ones(len(pd.inv_depth), np.uint8), obs=obs, imu=imu, prior=None, max_iterations=30)
    return out, truth
