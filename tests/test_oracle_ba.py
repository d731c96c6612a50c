"""CPU tests of the BA oracle (oracle/ba_oracle.cpp).  The reference pins none of this
(SURVEY.md section 8c), so the oracle is validated by: analytic-vs-finite-difference Jacobians
of every production factor (the method of the reference's own CostFunctionValidator,
estimation/ceres/cost_function_validator.h: central differences here, tolerance 2e-6 relative),
an independent numpy restatement of pre-integration, an independent numpy Schur complement
for marginalisation, and an independent minimiser (scipy) for the solver's fixed point."""
import os

import numpy as np
import pytest

from oracle import ba_oracle as bo
from tests import ba_synth as bs
from xrslam_amd import abi


def _fd(fun, s, n=15, eps=1e-7):
    f0 = fun(s)
    J = np.zeros((len(f0), n))
    for i in range(n):
        d = np.zeros(15)
        d[i] = eps
        J[:, i] = (fun(bo.state_plus(s, d)) - fun(bo.state_plus(s, -d))) / (2 * eps)
    return J


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.fixture(scope="module")
def window():
    return bs.make_window(K=5, L=60, seed=3)


def test_reprojection_jacobians(window):
    pd, _ = window
    st = pd.frame_state
    for o in range(0, len(pd.obs_tgt), 7):
        ft, fr, l = pd.obs_tgt[o], pd.obs_ref[o], pd.obs_lm[o]
        zt, zr, d = pd.obs_z_tgt[o], pd.obs_z_ref[o], pd.inv_depth[l]
        r, Jt, Jr, Jl = bo.eval_reprojection(st[ft], st[fr], d, zt, zr, bs.CAM_EXT, bs.SQRT_INV_COV)
        ft_ = lambda s: bo.eval_reprojection(s, st[fr], d, zt, zr, bs.CAM_EXT, bs.SQRT_INV_COV, False)[0]
        fr_ = lambda s: bo.eval_reprojection(st[ft], s, d, zt, zr, bs.CAM_EXT, bs.SQRT_INV_COV, False)[0]
        fl_ = lambda x: bo.eval_reprojection(st[ft], st[fr], x, zt, zr, bs.CAM_EXT, bs.SQRT_INV_COV, False)[0]
        assert _rel(_fd(ft_, st[ft], 6), Jt) < 2e-6
        assert _rel(_fd(fr_, st[fr], 6), Jr) < 2e-6
        assert _rel((fl_(d + 1e-7) - fl_(d - 1e-7)) / 2e-7, Jl) < 2e-6


def test_reprojection_residual_is_zero_at_truth():
    pd, truth = bs.make_window(K=4, L=30, seed=4, pixel_noise=0.0)
    st = truth["states"]
    for o in range(len(pd.obs_tgt)):
        r = bo.eval_reprojection(st[pd.obs_tgt[o]], st[pd.obs_ref[o]], truth["inv_depth"][pd.obs_lm[o]],
                                 pd.obs_z_tgt[o], pd.obs_z_ref[o], bs.CAM_EXT, bs.SQRT_INV_COV, False)[0]
        assert np.abs(r).max() < 1e-8


def test_rotation_prior_jacobian(window):
    pd, _ = window
    st = pd.frame_state
    for o in range(0, len(pd.obs_tgt), 11):
        ft, fr = pd.obs_tgt[o], pd.obs_ref[o]
        r, Jq = bo.eval_rotation(st[ft], st[fr], pd.obs_z_tgt[o], pd.obs_z_ref[o], bs.CAM_EXT, bs.SQRT_INV_COV)
        f = lambda s: bo.eval_rotation(s, st[fr], pd.obs_z_tgt[o], pd.obs_z_ref[o], bs.CAM_EXT, bs.SQRT_INV_COV,
                                       False)[0]
        assert _rel(_fd(f, st[ft], 3), Jq) < 2e-6


def test_imu_factor_jacobians(window):
    pd, _ = window
    st = pd.frame_state
    imu_ext = np.array([0.01, -0.02, 0.03, 0.9993, 0.02, -0.01, 0.03])
    imu_ext[:4] /= np.linalg.norm(imu_ext[:4])
    for ext in (bs.IMU_EXT, imu_ext):
        for k in range(len(pd.imu_i)):
            i, j = pd.imu_i[k], pd.imu_j[k]
            ref = st[i, 10:16] + np.array([1e-4, -2e-4, 1e-4, 3e-3, -2e-3, 1e-3])   # dbg, dba != 0
            r, Ji, Jj = bo.eval_imu(st[i], st[j], pd.imu_data[k], ref, ext)
            fi = lambda s: bo.eval_imu(s, st[j], pd.imu_data[k], ref, ext, False)[0]
            fj = lambda s: bo.eval_imu(st[i], s, pd.imu_data[k], ref, ext, False)[0]
            assert _rel(_fd(fi, st[i]), Ji) < 2e-6
            assert _rel(_fd(fj, st[j]), Jj) < 2e-6


def _preint_numpy(samples, t_end, bg, ba, noise36):
    """Independent numpy restatement of PreIntegrator::integrate (preintegrator.cpp:22-100)."""
    cw, ca, cbg, cba = [noise36[9 * i:9 * i + 9].reshape(3, 3) for i in range(4)]

    def hat(w):
        return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])

    def rjac(w):
        a = np.linalg.norm(w)
        H = hat(w)
        if a < 1e-7:
            return np.eye(3) - 0.5 * H + H @ H / 6
        return np.eye(3) - (1 - np.cos(a)) / a ** 2 * H + (a - np.sin(a)) / a ** 3 * H @ H

    q = np.array([0, 0, 0, 1.0])
    p = np.zeros(3)
    v = np.zeros(3)
    cov = np.zeros((15, 15))
    dq_dbg = np.zeros((3, 3))
    dp_dbg = np.zeros((3, 3))
    dp_dba = np.zeros((3, 3))
    dv_dbg = np.zeros((3, 3))
    dv_dba = np.zeros((3, 3))
    T = 0.0
    ts = list(samples[1:, 0]) + [t_end]
    for s, t1 in zip(samples, ts):
        dt = t1 - s[0]
        w = s[1:4] - bg
        a = s[4:7] - ba
        R = bs.qmat(q)
        E = bs.qmat(bs.qconj(bs.qexp(w * dt)))
        A = np.eye(9)
        A[0:3, 0:3] = E
        A[6:9, 0:3] = -dt * R @ hat(a)
        A[3:6, 0:3] = -0.5 * dt * dt * R @ hat(a)
        A[3:6, 6:9] = dt * np.eye(3)
        B = np.zeros((9, 6))
        B[0:3, 0:3] = dt * rjac(w * dt)
        B[6:9, 3:6] = dt * R
        B[3:6, 3:6] = 0.5 * dt * dt * R
        Q = np.zeros((6, 6))
        Q[0:3, 0:3] = cw / max(dt, 1e-7)
        Q[3:6, 3:6] = ca / max(dt, 1e-7)
        cov[0:9, 0:9] = A @ cov[0:9, 0:9] @ A.T + B @ Q @ B.T
        cov[9:12, 9:12] += cbg * dt
        cov[12:15, 12:15] += cba * dt
        dp_dbg = dp_dbg + dt * dv_dbg - 0.5 * dt * dt * R @ hat(a) @ dq_dbg
        dp_dba = dp_dba + dt * dv_dba - 0.5 * dt * dt * R
        dv_dbg = dv_dbg - dt * R @ hat(a) @ dq_dbg
        dv_dba = dv_dba - dt * R
        dq_dbg = E @ dq_dbg - dt * rjac(w * dt)
        T += dt
        p = p + dt * v + 0.5 * dt * dt * (R @ a)
        v = v + dt * (R @ a)
        q = bs.qmul(q, bs.qexp(w * dt))
        q = q / np.linalg.norm(q)
    return T, q, p, v, cov, (dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba)


def test_preintegration_matches_independent_numpy(window):
    pd, truth = window
    smp = truth["samples"][1]
    bg, ba = pd.frame_state[1, 10:13], pd.frame_state[1, 13:16]
    t_end = truth["times"][2]
    out = bo.preintegrate(smp, t_end, bg, ba, bs.NOISE36)
    T, q, p, v, cov, J = _preint_numpy(smp, t_end, bg, ba, bs.NOISE36)
    assert abs(out[0] - T) < 1e-12
    np.testing.assert_allclose(out[1:5], q, atol=1e-12)
    np.testing.assert_allclose(out[5:8], p, atol=1e-12)
    np.testing.assert_allclose(out[8:11], v, atol=1e-12)
    for k in range(5):
        np.testing.assert_allclose(out[11 + 9 * k:20 + 9 * k].reshape(3, 3), J[k], atol=1e-10)
    np.testing.assert_allclose(bo.preintegrate_cov(smp, t_end, bg, ba, bs.NOISE36), cov, rtol=1e-9, atol=1e-18)
    U = out[56:].reshape(15, 15)          # sqrt_inv_cov = LLT(cov^-1).L^T  =>  U^T U = cov^-1
    assert np.allclose(U, np.triu(U))
    np.testing.assert_allclose(U.T @ U @ cov, np.eye(15), atol=1e-6)


def test_preintegration_constant_rate_closed_form():
    w = np.array([0.3, -0.2, 0.5])
    a = np.array([0.0, 0.0, 0.0])
    n, dt = 40, 0.005
    smp = np.array([[i * dt, *w, *a] for i in range(n)])
    out = bo.preintegrate(smp, n * dt, np.zeros(3), np.zeros(3), bs.NOISE36, True, False)
    np.testing.assert_allclose(out[1:5], bs.qexp(w * n * dt), atol=1e-12)   # product of exp(w dt) is exact
    assert np.abs(out[5:11]).max() == 0
    # prediction with gravity only: free fall
    s0 = np.zeros(16)
    s0[3] = 1
    s1 = bo.predict(s0, out)
    T = n * dt
    np.testing.assert_allclose(s1[4:7], [0, 0, -0.5 * bs.GRAVITY * T * T], atol=1e-12)
    np.testing.assert_allclose(s1[7:10], [0, 0, -bs.GRAVITY * T], atol=1e-12)


def _numpy_marginal(md):
    """Independent dense assembly + Schur complements for the marginalisation (no robust loss)."""
    K = len(md.frame_state)
    N = 15 * K
    st = md.frame_state
    fidx = [i if i < md.victim else (i - 1 if i > md.victim else K - 1) for i in range(K)]
    L = len(md.inv_depth)
    H = np.zeros((N + L, N + L))
    b = np.zeros(N + L)
    # prior
    n = 15 * len(md.prior_frames)
    if n:
        delta = np.zeros(n)
        Bm = np.eye(n)
        for i, f in enumerate(md.prior_frames):
            lin = md.prior_lin[i]
            rq = bs.qlog(bs.qmul(bs.qconj(lin[0:4]), st[f, 0:4]))
            delta[15 * i:15 * i + 3] = rq
            delta[15 * i + 3:15 * i + 15] = st[f, 4:16] - lin[4:16]
            a = np.linalg.norm(rq)
            Hh = np.array([[0, -rq[2], rq[1]], [rq[2], 0, -rq[0]], [-rq[1], rq[0], 0]])
            Jr = np.eye(3) - 0.5 * Hh + Hh @ Hh / 6 if a < 1e-7 else (
                np.eye(3) - (1 - np.cos(a)) / a ** 2 * Hh + (a - np.sin(a)) / a ** 3 * Hh @ Hh)
            Bm[15 * i:15 * i + 3, 15 * i:15 * i + 3] = np.linalg.inv(Jr)
        r = md.prior_sqrt_info @ delta + md.prior_infovec
        J = md.prior_sqrt_info @ Bm
        cols = np.concatenate([15 * fidx[f] + np.arange(15) for f in md.prior_frames])
        H[np.ix_(cols, cols)] += J.T @ J
        b[cols] += J.T @ r
    for k in range(len(md.imu_i)):
        i, j = md.imu_i[k], md.imu_j[k]
        r, Ji, Jj = bo.eval_imu(st[i], st[j], md.imu_data[k], st[i, 10:16], md.imu_ext)
        J = np.zeros((15, N + L))
        J[:, 15 * fidx[i]:15 * fidx[i] + 15] = Ji
        J[:, 15 * fidx[j]:15 * fidx[j] + 15] = Jj
        H += J.T @ J
        b += J.T @ r
    for o in range(len(md.obs_tgt)):
        ft, fr, l = md.obs_tgt[o], md.obs_ref[o], md.obs_lm[o]
        r, Jt, Jr, Jl = bo.eval_reprojection(st[ft], st[fr], md.inv_depth[l], md.obs_z_tgt[o], md.obs_z_ref[o],
                                             md.cam_ext, md.sqrt_inv_cov)
        J = np.zeros((2, N + L))
        J[:, 15 * fidx[ft]:15 * fidx[ft] + 6] += Jt
        J[:, 15 * fidx[fr]:15 * fidx[fr] + 6] += Jr
        J[:, N + l] = Jl
        H += J.T @ J
        b += J.T @ r
    used = np.array([H[N + l, N + l] > 0 for l in range(L)])
    keep = np.concatenate([np.arange(N), N + np.where(used)[0]])
    H = H[np.ix_(keep, keep)]
    b = b[keep]
    R = N - 15
    elim = np.arange(R, len(keep))
    rem = np.arange(R)
    Hee = H[np.ix_(elim, elim)]
    Hre = H[np.ix_(rem, elim)]
    S = H[np.ix_(rem, rem)] - Hre @ np.linalg.solve(Hee, Hre.T)
    bs_ = b[rem] - Hre @ np.linalg.solve(Hee, b[elim])
    return S, bs_


def _marg_from_window(pd, victim=0):
    K = len(pd.frame_state)
    seen = set(pd.obs_lm[(pd.obs_ref == victim) | (pd.obs_tgt == victim)])
    sel = np.array([l in seen for l in pd.obs_lm])
    obs = dict(tgt=pd.obs_tgt[sel], ref=pd.obs_ref[sel], lm=pd.obs_lm[sel], z_tgt=pd.obs_z_tgt[sel],
               z_ref=pd.obs_z_ref[sel])
    ki = np.array([k for k in range(len(pd.imu_i)) if pd.imu_i[k] == victim or pd.imu_j[k] == victim], int)
    imu = dict(i=pd.imu_i[ki], j=pd.imu_j[ki], data=pd.imu_data[ki])
    prior = dict(frames=pd.prior_frames, sqrt_info=pd.prior_sqrt_info, infovec=pd.prior_infovec, lin=pd.prior_lin)
    return abi.MargProblemData(pd.frame_state, victim, pd.cam_ext, pd.imu_ext, pd.sqrt_inv_cov, prior, imu,
                               pd.inv_depth, obs)


def test_marginalization_matches_independent_schur():
    pd, truth = bs.make_window(K=6, L=80, seed=6)
    # move away from the prior's linearisation point so the prior residual/Jacobian are non-trivial,
    # but keep frame 0's pose (gauge prior 1e15) where it was
    pd.frame_state[1:, 4:7] += 1e-3
    md = _marg_from_window(pd, 0)
    si, iv, lin = bo.marginalize(md)
    S, b = _numpy_marginal(md)
    Lam = si.T @ si
    eta = si.T @ iv
    scale = np.abs(S).max()
    assert np.abs(Lam - S).max() / scale < 1e-9
    # sqrt_info^T infovec reproduces the information vector on the retained eigen-space
    w, V = np.linalg.eigh(S)
    P = V[:, w > 1e-8] @ V[:, w > 1e-8].T
    assert np.abs(eta - P @ b).max() / max(1.0, np.abs(b).max()) < 1e-7
    np.testing.assert_array_equal(lin, pd.frame_state[1:])


def test_marginalized_prior_keeps_the_optimum():
    """Marginalise-then-solve == solve-then-drop at the linearisation point: after marginalising frame 0
    at the full problem's optimum, the prior's gradient contribution must vanish there (b ~ 0 => infovec ~ 0
    up to the robust-loss reweighting the marginalisation deliberately ignores)."""
    pd, _ = bs.make_window(K=5, L=60, seed=8, pixel_noise=0.0, state_noise=0.0)
    md = _marg_from_window(pd, 0)
    si, iv, lin = bo.marginalize(md)
    # zero-noise data at the true states: every residual is ~0, so is the information vector
    assert np.abs(si.T @ iv).max() < 1e-3 * np.abs(si.T @ si).max()


def test_solver_reaches_the_same_minimum_as_scipy():
    scipy_opt = pytest.importorskip("scipy.optimize")
    pd, truth = bs.make_localize(seed=2)
    j = len(pd.frame_state) - 1
    ref = pd.copy()
    sm = bo.solve(pd)
    assert sm.usable and sm.termination == abi_conv() and sm.final_cost < sm.initial_cost

    def residuals(d15):
        s = bo.state_plus(ref.frame_state[j], d15)
        out = []
        for o in range(len(ref.obs_tgt)):
            r = bo.eval_reprojection(s, ref.frame_state[ref.obs_ref[o]], ref.inv_depth[ref.obs_lm[o]],
                                     ref.obs_z_tgt[o], ref.obs_z_ref[o], bs.CAM_EXT, bs.SQRT_INV_COV, False)[0]
            n2 = r @ r
            out.append(r * np.sqrt(np.log1p(n2) / max(n2, 1e-300)))     # 0.5*sum rho(|r|^2), rho = log(1+s)
        i = ref.imu_i[0]
        out.append(bo.eval_imu(ref.frame_state[i], s, ref.imu_data[0], ref.frame_state[i, 10:16], bs.IMU_EXT, False)[0])
        return np.concatenate(out)

    sol = scipy_opt.least_squares(residuals, np.zeros(15), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    s_ref = bo.state_plus(ref.frame_state[j], sol.x)
    assert abs(0.5 * sol.fun @ sol.fun - sm.final_cost) < 1e-5 * sm.final_cost
    np.testing.assert_allclose(pd.frame_state[j, 4:7], s_ref[4:7], atol=2e-5)
    np.testing.assert_allclose(pd.frame_state[j, 0:4], s_ref[0:4], atol=2e-5)


def abi_conv():
    return 0   # XRHIP_BA_CONVERGENCE


def test_solver_on_window_reduces_cost_and_respects_gauge():
    pd, truth = bs.make_window(K=8, L=120, seed=9)
    x0 = pd.frame_state.copy()
    sm = bo.solve(pd)
    assert sm.usable and sm.final_cost < 0.05 * sm.initial_cost
    assert 1 <= sm.iterations <= 30
    # gauge prior (1e15) pins the first pose
    np.testing.assert_allclose(pd.frame_state[0, :7], x0[0, :7], atol=1e-9)
    assert np.isfinite(pd.frame_state).all() and np.isfinite(pd.inv_depth).all()
    assert np.allclose(np.linalg.norm(pd.frame_state[:, :4], axis=1), 1.0, atol=1e-12)


def test_solver_trivial_and_fixed_problems():
    pd, _ = bs.make_window(K=4, L=30, seed=10)
    pd.frame_fix[:] = abi.FIX_POSE | abi.FIX_MOTION
    pd.landmark_fix[:] = 1
    before = pd.frame_state.copy()
    sm = bo.solve(pd)
    assert sm.iterations == 0 and sm.usable
    np.testing.assert_array_equal(pd.frame_state, before)


def test_se3_cost_function_pattern():
    """Mirror of the reference's only estimation unit test (xrslam-test/test/src/test_se3_cost_function.cpp:10-93):
    the toy factor r = p1 - (q*p2 + p) over a quaternion block with QuaternionParameterization
    (estimation/ceres/quaternion_parameterization.h:12-18: Plus = (q * expmap(dq)).normalized(), identity lift) and
    its hand-written Jacobians dr/ddq = R(q) hat(p2), dr/dp = -I, validated like CostFunctionValidator
    (cost_function_validator.h:22-26,289-367: forward differences through Plus, fd_epsilon 1e-9, tolerance 2e-6) before
    and after a solve.  The quaternion Plus is the oracle's (orc_state_plus), so the test pins the oracle's
    right-multiplicative update convention to the one the reference's test checks.  Inputs are seeded (the reference
    draws them from an unseeded std::random_device, so it holds no golden numbers)."""
    rng = np.random.default_rng(20240924)
    hat = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    def plus(q, p, d6):
        st = np.zeros(16)
        st[0:4], st[4:7] = q, p
        d = np.zeros(15)
        d[0:3], d[3:6] = d6[0:3], d6[3:6]
        out = bo.state_plus(st, d)
        return out[0:4], out[4:7]

    def validate(q, p, p1, p2):
        r0 = p1 - (bs.qrot(q, p2) + p)
        J = np.hstack([bs.qmat(q) @ hat(p2), -np.eye(3)])
        Jfd = np.zeros((3, 6))
        for i in range(6):
            d = np.zeros(6)
            d[i] = 1.0e-9
            qi, pi = plus(q, p, d)
            Jfd[:, i] = ((p1 - (bs.qrot(qi, p2) + pi)) - r0) / 1.0e-9
        return np.abs(J - Jfd).max(), r0, J

    for _ in range(20):
        p2 = rng.uniform(-1, 1, 3)
        ax, ay, az = rng.uniform(-1, 1, 3) * np.pi
        q = bs.qmul(bs.qmul(bs.qexp(np.array([ax, 0, 0])), bs.qexp(np.array([0, ay, 0]))), bs.qexp(np.array([0, 0, az])))
        p = rng.uniform(-1, 1, 3)
        p1 = bs.qrot(q, p2) + p + rng.uniform(-1, 1, 3) * 0.1
        err, r, J = validate(q, p, p1, p2)
        assert err < 2.0e-6
        for _it in range(50):       # Gauss-Newton on the 3-residual / 6-dof problem (minimum-norm step)
            q, p = plus(q, p, np.linalg.lstsq(J, -r, rcond=None)[0])
            err, r, J = validate(q, p, p1, p2)
            if np.abs(r).max() < 1e-14:
                break
        assert np.abs(r).max() < 1e-12          # an exact fit exists (6 dof, 3 residuals)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-15
        assert err < 2.0e-6


def test_pipeline_snapshots_regression():
    """Regression pin of the oracle's minimiser on the frozen problems of tests/golden/ba_snapshots (SURVEY.md 8d "S4";
    tests/golden/make_ba_snapshots.py): iteration count, accepted steps, termination, the per-iteration record (costs,
    model cost change, radius, accept / reject) and the final states must stay what they were when the fixtures were
    made.  The reference holds no golden vectors for this boundary (parity unpinned); this pins the
    oracle to ITSELF so that an edit which shifts a decision cannot pass unnoticed just because the GPU follows it."""
    from tests import ba_snapshots
    snaps = ba_snapshots.load_all()
    assert len(snaps) >= 5
    kinds = set()
    for name, pd, exp in snaps:
        kinds.add(name.split("_")[1])
        a = pd.copy()
        sm, trace = bo.solve_trace(a)
        assert sm.iterations == int(exp["iterations"]) and sm.successful_steps == int(exp["successful_steps"]), name
        assert sm.termination == int(exp["termination"]), name
        np.testing.assert_allclose(sm.initial_cost, float(exp["initial_cost"]), rtol=1e-12, err_msg=name)
        np.testing.assert_allclose(sm.final_cost, float(exp["final_cost"]), rtol=1e-10, err_msg=name)
        assert trace.shape == exp["trace"].shape, name
        np.testing.assert_array_equal(trace[:, [0, 8]], exp["trace"][:, [0, 8]], err_msg=name)        # iteration, accepted
        np.testing.assert_allclose(trace[:, [1, 2, 5, 7]], exp["trace"][:, [1, 2, 5, 7]], rtol=1e-9, err_msg=name)   # costs, radius, mu
        np.testing.assert_allclose(a.frame_state, exp["frame_state"], rtol=1e-9, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(a.inv_depth, exp["inv_depth"], rtol=1e-9, atol=1e-12, err_msg=name)
    assert kinds == {"localize", "subwindow", "window"}


def test_preintegration_queued_behind_a_solve_cpu_shim():
    """xrhip_ba_preintegrate_after_solve through the CPU shim (oracle/xrhip_shim.cpp exports the plug point's symbols over the
    oracle): the record equals the one integrated from the biases the solve returned -- what tests/test_ba_gpu.py asserts of
    the device path, where the batch runs behind the solve's last kernel."""
    import ctypes
    from tests import ba_snapshots, ba_synth as bs
    from xrslam_amd import ba
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libxrslam_oracle.so")
    if not os.path.exists(shim):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.dirname(shim))])
    ctx = ba.BaContext(lib=ba.configure(ctypes.CDLL(shim)))
    _, truth = bs.make_window(K=5, L=20, seed=31)
    smp, t_end = truth["samples"][1], truth["times"][2]
    for name, pd, _exp in ba_snapshots.load_all()[:3]:
        free = [f for f in range(len(pd.frame_state)) if (pd.frame_fix[f] & 2) == 0]
        f = free[-1] if free else 0
        b = pd.copy()
        ctx.preintegrate_after_solve(smp, t_end, f, bs.NOISE36)
        ctx.solve(b)
        chained = ctx.preintegrate_end()
        direct = ctx.preintegrate(smp, t_end, b.frame_state[f, 10:13], b.frame_state[f, 13:16], bs.NOISE36)
        np.testing.assert_array_equal(chained, direct, err_msg=name)
    ctx.preintegrate_after_solve(smp, t_end, 0, bs.NOISE36)
    with pytest.raises(Exception):
        ctx.preintegrate_end()
    name, pd, _exp = ba_snapshots.load_all()[0]
    ctx.preintegrate_after_solve(smp, t_end, len(pd.frame_state), bs.NOISE36)     # one past the last frame of the problem
    b = pd.copy()
    with pytest.raises(Exception):
        ctx.solve(b)                                                                # refused before the solve touches anything
    np.testing.assert_array_equal(b.frame_state, pd.frame_state)
    ctx.solve(b)                                                                    # and the context is usable again
    ctx.close()


def test_overlapped_solve_cpu_shim():
    """xrhip_ba_solve_overlapped through the CPU shim: the caller's host work runs exactly once and the solve's results are those of
    xrhip_ba_solve (the device path's half of this is tests/test_ba_gpu.py::test_overlapped_solve_runs_the_host_work_once_...)."""
    import ctypes
    from tests import ba_snapshots
    from xrslam_amd import ba
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libxrslam_oracle.so")
    if not os.path.exists(shim):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.dirname(shim))])
    ctx = ba.BaContext(lib=ba.configure(ctypes.CDLL(shim)))
    for name, pd, _exp in ba_snapshots.load_all()[:2]:
        a, b = pd.copy(), pd.copy()
        sa = ctx.solve(a)
        calls = []
        sb = ctx.solve(b, host_work=lambda: calls.append(1))
        assert calls == [1], name
        np.testing.assert_array_equal(a.frame_state, b.frame_state, err_msg=name)
        np.testing.assert_array_equal(a.inv_depth, b.inv_depth, err_msg=name)
        assert (sa.iterations, sa.termination, sa.final_cost) == (sb.iterations, sb.termination, sb.final_cost)
    ctx.close()
