"""GPU parity tests of the bundle adjustment: HIP path (C ABI) vs the CPU oracle on identical
problems.  Floating point: 1e-4 relative per north_star is the bar; the observed agreement is
~1e-9 and is asserted at 1e-7 so regressions in summation order or math show up."""
import os

import numpy as np
import pytest

from tests import ba_synth as bs
from xrslam_amd import abi

pytestmark = pytest.mark.gpu

DUMP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


@pytest.fixture(scope="module")
def ctx():
    from xrslam_amd import ba
    return ba.BaContext()


@pytest.fixture(scope="module")
def bo():
    from oracle import ba_oracle
    return ba_oracle


def _dump(name, **arrs):
    try:
        os.makedirs(DUMP, exist_ok=True)
        np.savez_compressed(os.path.join(DUMP, name + ".npz"), **arrs)
    except Exception:
        pass


def test_mfma_schur_product_asymmetric(ctx):
    rng = np.random.RandomState(0)
    for Ln, P in ((150, 60), (37, 66), (600, 120), (5, 6)):
        W = rng.randn(Ln, P) * (1 + np.arange(P))[None, :]     # asymmetric in the columns
        w = rng.rand(Ln) + 0.1
        T = ctx.debug_schur(W, w)
        ref = W.T @ (w[:, None] * W)
        assert np.abs(T - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.abs(T - T.T).max() <= 1e-12 * np.abs(ref).max()


def _compare_linearization(ctx, bo, pd, tag):
    cost_o, H_o, g_o, po, mo, lo = bo.linearize(pd)
    dev = ctx.debug_linearize(pd)
    F, Ln = len(pd.frame_state), len(pd.inv_depth)
    n_local = H_o.shape[0]
    # map the oracle's compact local layout to frame-major + landmarks
    idx = -np.ones(15 * F + Ln, int)
    for f in range(F):
        if po[f] >= 0:
            idx[15 * f:15 * f + 6] = po[f] + np.arange(6)
        if mo[f] >= 0:
            idx[15 * f + 6:15 * f + 15] = mo[f] + np.arange(9)
    for l in range(Ln):
        if lo[l] >= 0:
            idx[15 * F + l] = lo[l]
    full = np.zeros((15 * F + Ln, 15 * F + Ln))
    gfull = np.zeros(15 * F + Ln)
    act = idx >= 0
    full[np.ix_(act, act)] = H_o[np.ix_(idx[act], idx[act])]
    gfull[act] = g_o[idx[act]]
    n = 15 * F
    Hd = np.zeros_like(full)
    Hd[:n, :n] = dev["H"]
    pose_cols = np.concatenate([15 * f + np.arange(6) for f in range(F)])
    for l in range(Ln):
        Hd[n + l, n + l] = dev["hll"][l] if act[n + l] else 0.0
        Hd[pose_cols, n + l] = dev["W"][l]
        Hd[n + l, pose_cols] = dev["W"][l]
    gd = np.concatenate([dev["g"], np.where(act[n:], dev["gl"], 0.0)])
    scale = np.abs(full).max()
    ok = (abs(dev["cost"] - cost_o) <= 1e-10 * max(1.0, abs(cost_o)) and np.abs(Hd - full).max() <= 1e-9 * scale and
          np.abs(gd - gfull).max() <= 1e-9 * max(1.0, np.abs(gfull).max()))
    if not ok:
        _dump("ba_lin_mismatch_" + tag, Hd=Hd, Ho=full, gd=gd, go=gfull, cost=np.array([dev["cost"], cost_o]))
    assert abs(dev["cost"] - cost_o) <= 1e-10 * max(1.0, abs(cost_o))
    # 1e15 gauge prior entries (1e30 in H) need a relative comparison per element
    denom = np.maximum(np.abs(full), 1e-6 * np.sqrt(np.outer(np.abs(np.diag(full)) + 1e-300, np.abs(np.diag(full)) + 1e-300)))
    assert (np.abs(Hd - full) / np.maximum(denom, 1e-300)).max() < 1e-6
    assert np.abs(gd - gfull).max() <= 1e-8 * max(1.0, np.abs(gfull).max())


def test_linearization_matches_oracle(ctx, bo):
    pd, _ = bs.make_window(K=10, L=150, seed=1)
    _compare_linearization(ctx, bo, pd, "window")
    pd, _ = bs.make_localize(seed=2)
    _compare_linearization(ctx, bo, pd, "localize")
    pd, _ = bs.make_window(K=6, L=80, seed=5, n_fixed_first=2)
    _compare_linearization(ctx, bo, pd, "fixed2")


def _solve_both(ctx, bo, pd, tag, rtol=1e-7):
    a, b = pd.copy(), pd.copy()
    sm_o = bo.solve(a)
    sm_h = ctx.solve(b)
    good = (sm_o.iterations == sm_h.iterations and sm_o.termination == sm_h.termination and
            np.allclose(a.frame_state, b.frame_state, rtol=rtol, atol=1e-9) and
            np.allclose(a.inv_depth, b.inv_depth, rtol=rtol, atol=1e-9))
    if not good:
        _dump("ba_solve_mismatch_" + tag, so=a.frame_state, sh=b.frame_state, do=a.inv_depth, dh=b.inv_depth,
              meta=np.array([sm_o.iterations, sm_h.iterations, sm_o.termination, sm_h.termination,
                             sm_o.successful_steps, sm_h.successful_steps, sm_o.final_cost, sm_h.final_cost,
                             sm_o.initial_cost, sm_h.initial_cost]))
    assert abs(sm_h.initial_cost - sm_o.initial_cost) <= 1e-9 * sm_o.initial_cost
    assert sm_h.iterations == sm_o.iterations and sm_h.successful_steps == sm_o.successful_steps
    assert sm_h.termination == sm_o.termination and sm_h.usable == sm_o.usable
    # north_star tolerance: 1e-4 relative on pose/velocity/bias states; assert much tighter
    np.testing.assert_allclose(b.frame_state, a.frame_state, rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(b.inv_depth, a.inv_depth, rtol=rtol, atol=1e-9)
    assert abs(sm_h.final_cost - sm_o.final_cost) <= 1e-8 * sm_o.final_cost
    return sm_o, sm_h


# K = 11: 165 unknowns, the tiled LDS Cholesky (dense_lds.hip.h tl_*); K = 12: 180 unknowns no longer fit the tile layout and take the
# packed triangle in LDS (the window holds 12 frames when manage_keyframe promoted a subframe and the new frame in one step)
@pytest.mark.parametrize("K,Ln,seed", [(10, 150, 1), (11, 150, 2), (6, 60, 3), (12, 150, 4)])
def test_refine_window_solve_parity(ctx, bo, K, Ln, seed):
    pd, _ = bs.make_window(K=K, L=Ln, seed=seed)
    sm_o, sm_h = _solve_both(ctx, bo, pd, "window%d" % seed)
    assert sm_h.final_cost < sm_h.initial_cost


def test_localize_solve_parity(ctx, bo):
    pd, truth = bs.make_localize(seed=2)
    sm_o, sm_h = _solve_both(ctx, bo, pd, "localize")
    assert sm_h.termination == 0 and sm_h.successful_steps >= 2


def test_subwindow_sized_problem_parity(ctx, bo):
    """refine_subwindow: a fixed keyframe, a few free subframes, reprojection priors only (no free landmark:
    the Schur complement is empty)."""
    pd, _ = bs.make_window(K=4, L=60, seed=31, with_prior=False, n_fixed_first=1)
    pd.landmark_fix[:] = 1
    _solve_both(ctx, bo, pd, "subwindow")


def test_vision_only_and_fixed_frames_parity(ctx, bo):
    pd, _ = bs.make_window(K=6, L=80, seed=5)
    pd.frame_fix[:] = abi.FIX_MOTION
    pd.imu_i, pd.imu_j, pd.imu_data = pd.imu_i[:0], pd.imu_j[:0], pd.imu_data[:0]
    _solve_both(ctx, bo, pd, "vision_only")
    pd, _ = bs.make_window(K=7, L=80, seed=6, n_fixed_first=1, with_prior=False)
    _solve_both(ctx, bo, pd, "first_fixed_noprior")


def test_rotation_prior_factors_parity(ctx, bo):
    """refine_subwindow's rotation branch: rotation prior factors on the last frame (ceres/rotation_factor.h)."""
    pd, _ = bs.make_window(K=4, L=60, seed=7, with_prior=False, n_fixed_first=1)
    last = len(pd.frame_state) - 1
    keep = pd.obs_tgt == last
    rot = dict(tgt=pd.obs_tgt[keep], ref=pd.obs_ref[keep], z_tgt=pd.obs_z_tgt[keep], z_ref=pd.obs_z_ref[keep])
    obs = dict(tgt=pd.obs_tgt[~keep], ref=pd.obs_ref[~keep], lm=pd.obs_lm[~keep], z_tgt=pd.obs_z_tgt[~keep],
               z_ref=pd.obs_z_ref[~keep])
    imu = dict(i=pd.imu_i, j=pd.imu_j, data=pd.imu_data)
    p2 = abi.BaProblemData(pd.frame_state, pd.frame_fix, pd.cam_ext, pd.imu_ext, pd.sqrt_inv_cov, pd.inv_depth,
                           np.ones(len(pd.inv_depth), np.uint8), obs=obs, rot=rot, imu=imu)
    _compare_linearization(ctx, bo, p2, "rot")
    _solve_both(ctx, bo, p2, "rot")


def _stiff_prior(pd, rng, residual_scale):
    """a dense upper-triangular sqrt_info whose rows span six decades (bias / velocity rows of a long-lived prior), linearised a
    millimetre away from the current state, with an infovec that leaves only `residual_scale` of S delta standing: the state is close
    to the prior's minimiser, S delta and infovec cancel"""
    K = len(pd.frame_state)
    n = 15 * (K - 1)
    lin = pd.frame_state[:K - 1].copy()
    lin[:, 4:16] -= 1.0e-3 * rng.randn(K - 1, 12)
    delta = np.zeros((K - 1, 15))
    delta[:, 3:] = pd.frame_state[:K - 1, 4:16] - lin[:, 4:16]        # p, v, bg, ba: plain differences; rotations coincide
    S = np.triu(rng.randn(n, n)) * np.logspace(0, 6, n)[rng.permutation(n), None]
    S[np.arange(n), np.arange(n)] += np.sign(S[np.arange(n), np.arange(n)]) * np.abs(S).sum(1) * 0.05
    Sd = S @ delta.ravel()
    infovec = -Sd + residual_scale * np.abs(Sd).max() * rng.randn(n)
    return dict(frames=np.arange(K - 1), sqrt_info=S, infovec=infovec, lin=lin), S, delta.ravel(), infovec


def test_prior_gradient_of_a_stiff_prior_near_its_minimum(ctx, bo):
    """ADVICE r3 (low, ba_kernels.hip.h: prior row blocks): the prior's gradient is formed as t = Lambda delta + c0 (Lambda = S^T S,
    c0 = S^T infovec, once per solve) instead of S^T (S delta + infovec).  Equal on paper; near the prior's minimum S delta ~ -infovec
    and the first form's rounding error is ~eps |S|^T |S| |delta| where the second's is ~eps |S|^T |r|.  This pins what that costs:
    (a) prior-only problem, gradient against the direct product in extended precision: within the first form's own forward bound,
    and the loss against the direct form stays below 1e-6 of the gradient's scale; (b) a window problem carrying such a prior:
    the trust-region record (iterations, accepted steps, termination) and the states are the oracle's, which forms S^T r."""
    rng = np.random.RandomState(7)
    pd, _ = bs.make_window(K=5, L=40, seed=11)
    prior, S, delta, infovec = _stiff_prior(pd, rng, 1.0e-7)
    K = len(pd.frame_state)
    only = abi.BaProblemData(pd.frame_state, np.zeros(K, np.uint8), bs.CAM_EXT, bs.IMU_EXT, bs.SQRT_INV_COV, np.zeros(0), None,
                             obs=None, imu=None, prior=prior)
    dev = ctx.debug_linearize(only)
    L_ = np.longdouble
    r = S.astype(L_) @ delta.astype(L_) + infovec.astype(L_)
    g_ref = np.asarray(S.astype(L_).T @ r, np.float64)
    g_dev = dev["g"][:15 * (K - 1)]
    sel = np.concatenate([15 * f + np.arange(3, 15) for f in range(K - 1)])      # p, v, bg, ba rows (identity Jacobians)
    eps = np.finfo(np.float64).eps
    n = S.shape[0]
    bound_lambda = 4 * n * eps * (np.abs(S).T @ (np.abs(S) @ np.abs(delta)) + np.abs(S).T @ np.abs(infovec))
    err = np.abs(g_dev - g_ref)[sel]
    assert np.all(err <= bound_lambda[sel]), (err.max(), bound_lambda[sel].min())
    scale = np.abs(g_ref[sel]).max()
    assert err.max() <= 1.0e-6 * scale, (err.max(), scale)
    np.testing.assert_allclose(dev["cost"], 0.5 * float(r @ r), rtol=1e-9)
    # (b) the same kind of prior inside a window solve
    pd2, _ = bs.make_window(K=6, L=60, seed=12)
    prior2, _, _, _ = _stiff_prior(pd2, rng, 1.0e-4)
    pd2._set_prior(prior2)
    sm_o, sm_h = _solve_both(ctx, bo, pd2, "stiff_prior", rtol=1e-6)
    assert sm_h.iterations >= 2


def test_trivial_and_invalid_problems(ctx):
    from xrslam_amd._lib import XrhipError
    pd, _ = bs.make_window(K=4, L=30, seed=10)
    pd.frame_fix[:] = abi.FIX_POSE | abi.FIX_MOTION
    pd.landmark_fix[:] = 1
    before = pd.frame_state.copy()
    sm = ctx.solve(pd)
    assert sm.iterations == 0 and sm.usable
    np.testing.assert_array_equal(pd.frame_state, before)
    pd, _ = bs.make_window(K=4, L=30, seed=10)
    pd.obs_tgt[0] = 99
    with pytest.raises(XrhipError):
        ctx.solve(pd)


def test_large_window_uses_global_cholesky_path(ctx, bo):
    """cfg-C sized window (15 KF): the reduced system exceeds the LDS budget -> global-memory Cholesky."""
    pd, _ = bs.make_window(K=16, L=300, seed=11)
    _solve_both(ctx, bo, pd, "k16")


def test_config5_sized_window(ctx, bo):
    """BASELINE config 5: 600 features, 20-keyframe window plus the new frame (315 frame unknowns, ~8000 observations)."""
    pd, _ = bs.make_window(K=21, L=600, seed=4)
    assert len(pd.obs_lm) > 6000
    sm_o, sm_h = _solve_both(ctx, bo, pd, "k21")
    assert sm_h.final_cost < sm_h.initial_cost


def test_pipeline_snapshots_match_golden(ctx):
    """SURVEY.md 8d "S4": the frozen problems the pipeline assembled on the S1 / S2 / S3 streams (localize_newframe,
    refine_subwindow, refine_window with prior; tests/golden/ba_snapshots, made by tests/golden/make_ba_snapshots.py)
    against the oracle's committed result -- iteration count, accepted steps, termination, costs and final states.  Unlike
    the oracle-vs-GPU tests above, the expected values here are files: an edit of the oracle that shifts an accept / reject
    decision shows up in tests/test_oracle_ba.py, an edit of the kernels here."""
    from tests import ba_snapshots
    snaps = ba_snapshots.load_all()
    assert len(snaps) >= 5
    for name, pd, exp in snaps:
        b = pd.copy()
        sm = ctx.solve(b)
        ok = (sm.iterations == int(exp["iterations"]) and sm.successful_steps == int(exp["successful_steps"]) and
              np.allclose(b.frame_state, exp["frame_state"], rtol=1e-6, atol=1e-9))
        if not ok:
            _dump("ba_snapshot_mismatch_" + name, sh=b.frame_state, so=exp["frame_state"], dh=b.inv_depth, do=exp["inv_depth"],
                  meta=np.array([sm.iterations, int(exp["iterations"]), sm.successful_steps, int(exp["successful_steps"]),
                                 sm.termination, int(exp["termination"]), sm.final_cost, float(exp["final_cost"])]))
        assert sm.iterations == int(exp["iterations"]), name
        assert sm.successful_steps == int(exp["successful_steps"]), name
        assert sm.termination == int(exp["termination"]), name
        assert abs(sm.initial_cost - float(exp["initial_cost"])) <= 1e-9 * float(exp["initial_cost"]), name
        assert abs(sm.final_cost - float(exp["final_cost"])) <= 1e-8 * float(exp["final_cost"]), name
        np.testing.assert_allclose(b.frame_state, exp["frame_state"], rtol=1e-6, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(b.inv_depth, exp["inv_depth"], rtol=1e-6, atol=1e-9, err_msg=name)


def test_precision_study_kernels(ctx):
    """csrc/study_api.hip (BASELINE config 5's fp32 / bf16 study): the f64 kernel equals the product's Schur kernel, the f32
    and bf16 kernels equal a numpy contraction of the equally rounded operands up to accumulation order."""
    from xrslam_amd import ba
    rng = np.random.RandomState(5)
    for Ln, P in ((600, 126), (150, 66), (70, 6)):
        W = rng.randn(Ln, P) * (1 + np.arange(P))[None, :]
        w = rng.rand(Ln) + 0.5
        T, ms = ba.study_schur_precision(W, w, reps=3)
        A = np.sqrt(w)[:, None] * W
        ref = A.T @ A
        scale = np.abs(ref).max()
        assert np.abs(T["f64"] - ref).max() <= 1e-12 * scale
        assert np.abs(T["f64"] - ctx.debug_schur(W, w)).max() <= 1e-12 * scale
        A32 = A.astype(np.float32)
        assert np.abs(T["f32"] - (A32.astype(np.float64).T @ A32.astype(np.float64))).max() <= 2e-5 * scale
        u = A32.view(np.uint32).astype(np.uint64)
        A16 = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).astype(np.float64)
        assert np.abs(T["bf16"] - A16.T @ A16).max() <= 2e-5 * scale          # same operands: only the f32 accumulation differs
        assert np.abs(T["bf16"] - ref).max() <= 2e-2 * scale                    # and bf16 itself is that coarse
        assert all(v > 0 for v in ms.values())


def test_preintegration_queued_behind_a_solve(ctx):
    """xrhip_ba_preintegrate_after_solve: the integration starts from the biases the solve leaves on the device.  Same
    kernel, same inputs as the host-mediated form (solve, read the biases back, xrhip_ba_preintegrate) -> the same bits, for
    the single-launch solves (the batch runs behind kb_chain, reading the device state) and for the window solves (launched
    when the solve returns).  A batch that no solve follows is refused by _end, and the context stays usable."""
    from tests import ba_snapshots
    _, truth = bs.make_window(K=5, L=20, seed=31)
    smp, t_end = truth["samples"][1], truth["times"][2]
    moved = 0
    for name, pd, _exp in ba_snapshots.load_all():
        free = [f for f in range(len(pd.frame_state)) if (pd.frame_fix[f] & 2) == 0]      # frames whose motion is optimised
        f = free[-1] if free else 0
        b = pd.copy()
        ctx.preintegrate_after_solve(smp, t_end, f, bs.NOISE36)
        ctx.solve(b)
        chained = ctx.preintegrate_end()
        direct = ctx.preintegrate(smp, t_end, b.frame_state[f, 10:13], b.frame_state[f, 13:16], bs.NOISE36)
        if not np.array_equal(chained, direct):
            _dump("preint_chained_mismatch_" + name, chained=chained, direct=direct, bias=b.frame_state[f, 10:16],
                  bias0=pd.frame_state[f, 10:16], f=np.array([f]))
        np.testing.assert_array_equal(chained, direct, err_msg="%s (frame %d)" % (name, f))
        moved += int(not np.array_equal(b.frame_state[f, 10:16], pd.frame_state[f, 10:16]))
    assert moved >= 2          # the biases the integrations started from are the solves' results, not their inputs
    ctx.preintegrate_after_solve(smp, t_end, 0, bs.NOISE36)
    with pytest.raises(Exception):
        ctx.preintegrate_end()
    name, pd, _exp = ba_snapshots.load_all()[0]
    ctx.preintegrate_after_solve(smp, t_end, len(pd.frame_state), bs.NOISE36)     # one past the last frame of the problem
    b = pd.copy()
    with pytest.raises(Exception):
        ctx.solve(b)                                                                # refused before anything is queued
    np.testing.assert_array_equal(b.frame_state, pd.frame_state)
    ctx.solve(b)                                                                    # and the context is usable again
    np.testing.assert_array_equal(ctx.preintegrate(smp, t_end, np.zeros(3), np.zeros(3), bs.NOISE36),
                                  ctx.preintegrate(smp, t_end, np.zeros(3), np.zeros(3), bs.NOISE36))


def test_marginalization_guard_bound_is_the_trace_bound(ctx, bo):
    """Round 5: km_chol's eigenvalue guard forms L^-1 in 16-wide blocks on the matrix cores (tri_inverse_blocked) instead of one forward
    substitution per thread.  The bound it hands the gate, lambda_min >= 1 / trace(A^-1), against numpy on the same marginal matrix
    (restricted to its support): equal to rounding and a rigorous lower bound of the smallest eigenvalue.  A window's FIRST marginalisation
    is rank deficient (no prior: the eigen path, as in the pipeline); the ones that follow carry the previous prior and take the fast path."""
    checked = 0
    for K, Ln, seed in ((11, 150, 21), (16, 300, 23)):
        pd, _ = bs.make_window(K=K, L=Ln, seed=seed)
        pd.frame_state[1:, 4:7] += 1e-3
        cur = pd
        for step in range(3):
            md = _marg_problem(cur, 0)
            si_h, iv_h, lin_h = ctx.marginalize(md)
            lam, st = ctx.marg_guard()
            A = si_h.T @ si_h
            sup = np.where(np.abs(A).sum(1) > 0)[0]
            assert st[0] == 0
            if not st[4]:                                              # the Cholesky fast path stood: its guard is what decided
                assert st[1] == len(sup)
                Ac = A[np.ix_(sup, sup)]
                ref = 1.0 / np.trace(np.linalg.inv(Ac))
                assert abs(lam - ref) <= 1e-5 * ref, (K, step, lam, ref)
                assert lam <= np.linalg.eigvalsh(Ac)[0] * (1 + 1e-6) and lam > 1e-8
                checked += 1
            # the next window: the victim gone, the prior just produced on the frames that remain
            n = len(cur.frame_state)
            prior = dict(frames=np.arange(n - 1), sqrt_info=si_h, infovec=iv_h, lin=lin_h)
            keep = (cur.obs_tgt > 0) & (cur.obs_ref > 0)
            obs = dict(tgt=cur.obs_tgt[keep] - 1, ref=cur.obs_ref[keep] - 1, lm=cur.obs_lm[keep], z_tgt=cur.obs_z_tgt[keep],
                       z_ref=cur.obs_z_ref[keep])
            ki = cur.imu_i > 0
            imu = dict(i=cur.imu_i[ki] - 1, j=cur.imu_j[ki] - 1, data=cur.imu_data[ki])
            cur = abi.BaProblemData(cur.frame_state[1:], cur.frame_fix[1:], cur.cam_ext, cur.imu_ext, cur.sqrt_inv_cov, cur.inv_depth,
                                    None, obs=obs, imu=imu, prior=prior)
    assert checked >= 2, "no marginalisation took the fast path"


def _marg_problem(pd, victim=0):
    seen = set(pd.obs_lm[(pd.obs_ref == victim) | (pd.obs_tgt == victim)])
    sel = np.array([l in seen for l in pd.obs_lm])
    obs = dict(tgt=pd.obs_tgt[sel], ref=pd.obs_ref[sel], lm=pd.obs_lm[sel], z_tgt=pd.obs_z_tgt[sel],
               z_ref=pd.obs_z_ref[sel])
    ki = np.array([k for k in range(len(pd.imu_i)) if pd.imu_i[k] == victim or pd.imu_j[k] == victim], int)
    imu = dict(i=pd.imu_i[ki], j=pd.imu_j[ki], data=pd.imu_data[ki])
    prior = dict(frames=pd.prior_frames, sqrt_info=pd.prior_sqrt_info, infovec=pd.prior_infovec, lin=pd.prior_lin)
    return abi.MargProblemData(pd.frame_state, victim, pd.cam_ext, pd.imu_ext, pd.sqrt_inv_cov, prior, imu,
                               pd.inv_depth, obs)


@pytest.mark.parametrize("K,Ln,seed", [(11, 150, 21), (6, 80, 22), (16, 300, 23), (21, 600, 24)])
def test_marginalization_parity(ctx, bo, K, Ln, seed):
    """sqrt_info / infovec are only defined up to an orthogonal transform of the eigenbasis, so parity is
    asserted on the invariants the solver consumes: Lambda = S^T S and eta = S^T infovec."""
    pd, _ = bs.make_window(K=K, L=Ln, seed=seed)
    pd.frame_state[1:, 4:7] += 1e-3
    md = _marg_problem(pd, 0)
    si_o, iv_o, lin_o = bo.marginalize(md)
    si_h, iv_h, lin_h = ctx.marginalize(md)
    Lo, Lh = si_o.T @ si_o, si_h.T @ si_h
    eo, eh = si_o.T @ iv_o, si_h.T @ iv_h
    ok = np.abs(Lh - Lo).max() <= 1e-8 * np.abs(Lo).max() and np.abs(eh - eo).max() <= 1e-7 * max(1.0, np.abs(eo).max())
    if not ok:
        _dump("marg_mismatch_%d" % seed, Lo=Lo, Lh=Lh, eo=eo, eh=eh)
    assert np.abs(Lh - Lo).max() <= 1e-8 * np.abs(Lo).max()
    assert np.abs(eh - eo).max() <= 1e-7 * max(1.0, np.abs(eo).max())
    np.testing.assert_array_equal(lin_h, lin_o)
    # second marginalisation on top of the first (prior with a dense sqrt_info), via a solve in between
    prior = dict(frames=np.arange(K - 1), sqrt_info=si_h, infovec=iv_h, lin=lin_h)
    keep = (pd.obs_tgt > 0) & (pd.obs_ref > 0)
    obs = dict(tgt=pd.obs_tgt[keep] - 1, ref=pd.obs_ref[keep] - 1, lm=pd.obs_lm[keep], z_tgt=pd.obs_z_tgt[keep],
               z_ref=pd.obs_z_ref[keep])
    ki = pd.imu_i > 0
    imu = dict(i=pd.imu_i[ki] - 1, j=pd.imu_j[ki] - 1, data=pd.imu_data[ki])
    p2 = abi.BaProblemData(pd.frame_state[1:], pd.frame_fix[1:], pd.cam_ext, pd.imu_ext, pd.sqrt_inv_cov, pd.inv_depth,
                           None, obs=obs, imu=imu, prior=prior)
    _solve_both(ctx, bo, p2, "after_marg%d" % seed, rtol=1e-6)


def test_marginalization_begin_end_equals_the_blocking_call(ctx):
    """The asynchronous pair queues the same kernels on the context's stream; solves on ANOTHER context may run in
    between (the pipeline gives the marginalisation a context of its own)."""
    from xrslam_amd import ba
    other = ba.BaContext()
    pd, _ = bs.make_window(K=11, L=150, seed=21)
    md = _marg_problem(pd)
    si0, iv0, lin0 = ctx.marginalize(md)
    ctx.marginalize_begin(md)
    pd2, _ = bs.make_localize(seed=2)
    sm = other.solve(pd2)                       # unrelated work on a second context while the first one is busy
    assert sm.usable
    si1, iv1, lin1 = ctx.marginalize_end()
    np.testing.assert_array_equal(si1, si0)
    np.testing.assert_array_equal(iv1, iv0)
    np.testing.assert_array_equal(lin1, lin0)
    with pytest.raises(Exception):
        ctx.marginalize_end()                   # nothing in flight any more


def test_preintegration_parity(ctx, bo):
    pd, truth = bs.make_window(K=5, L=20, seed=31)
    for k, smp in enumerate(truth["samples"]):
        t_end = truth["times"][k + 1]
        bg, ba = pd.frame_state[k, 10:13], pd.frame_state[k, 13:16]
        for jac, cov in ((True, True), (False, False)):
            o = bo.preintegrate(smp, t_end, bg, ba, bs.NOISE36, jac, cov)
            h = ctx.preintegrate(smp, t_end, bg, ba, bs.NOISE36, jac, cov)
            np.testing.assert_allclose(h[:11], o[:11], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(h[11:56], o[11:56], rtol=1e-10, atol=1e-13)
            if cov:
                U_o, U_h = o[56:].reshape(15, 15), h[56:].reshape(15, 15)
                assert np.abs(U_h - U_o).max() <= 1e-7 * np.abs(U_o).max()
                assert np.abs(U_h.T @ U_h - U_o.T @ U_o).max() <= 1e-8 * np.abs(U_o.T @ U_o).max()
    # a single sample and a long (subframe-compressed) segment
    one = truth["samples"][0][:1]
    # (one sample gives a rank-deficient covariance, so only the tracker's no-covariance form is defined)
    np.testing.assert_allclose(
        ctx.preintegrate(one, one[0, 0] + 0.005, np.zeros(3), np.zeros(3), bs.NOISE36, True, False)[:56],
        bo.preintegrate(one, one[0, 0] + 0.005, np.zeros(3), np.zeros(3), bs.NOISE36, True, False)[:56],
        rtol=1e-10, atol=1e-14)
    long = np.concatenate(truth["samples"][:3])
    np.testing.assert_allclose(ctx.preintegrate(long, truth["times"][3], bg, ba, bs.NOISE36)[:56],
                               bo.preintegrate(long, truth["times"][3], bg, ba, bs.NOISE36)[:56], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("n", [31, 32, 33, 64, 65, 100, 161])
def test_preintegration_parity_across_chunk_boundaries(ctx, bo, n):
    """kp_preintegrate works through a job in chunks of 32 samples and composes the chunks' maps (preint_compose): segment lengths
    around the chunk boundaries and several chunks long, with Jacobians and covariance, against the oracle's sample-by-sample loop."""
    rng = np.random.RandomState(100 + n)
    smp = np.zeros((n, 7))
    smp[:, 0] = 2.0 + 0.005 * np.arange(n)
    smp[:, 1:4] = 0.3 * rng.randn(n, 3) + np.array([0.2, -0.1, 0.4])
    smp[:, 4:7] = np.array([0.3, -0.2, 9.7]) + 0.5 * rng.randn(n, 3)
    t_end = float(smp[-1, 0] + 0.003)
    bg, ba = 1e-3 * rng.randn(3), 1e-2 * rng.randn(3)
    o = bo.preintegrate(smp, t_end, bg, ba, bs.NOISE36, True, True)
    h = ctx.preintegrate(smp, t_end, bg, ba, bs.NOISE36, True, True)
    np.testing.assert_allclose(h[:11], o[:11], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(h[11:56], o[11:56], rtol=1e-9, atol=1e-12)
    U_o, U_h = o[56:].reshape(15, 15), h[56:].reshape(15, 15)
    assert np.abs(U_h - U_o).max() <= 1e-7 * np.abs(U_o).max()
    assert np.abs(U_h.T @ U_h - U_o.T @ U_o).max() <= 1e-8 * np.abs(U_o.T @ U_o).max()
    # the covariance-free forms the tracker uses
    np.testing.assert_allclose(ctx.preintegrate(smp, t_end, bg, ba, bs.NOISE36, False, False)[:11], o[:11], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ctx.preintegrate(smp, t_end, bg, ba, bs.NOISE36, True, False)[:56], o[:56], rtol=1e-9, atol=1e-12)


def test_schur_precision_knob_is_a_study_aid_that_resets(ctx, bo):
    """xrhip_ba_debug_set_schur_precision (BASELINE config 5's study): f32 operands move the result of a window solve by ~1e-7, and
    mode 0 afterwards is the product's f64 path again -- bit for bit."""
    pd, _ = bs.make_window(K=8, L=100, seed=41)
    a, b, c2 = pd.copy(), pd.copy(), pd.copy()
    ctx.solve(a)
    ctx.set_schur_precision(1)
    try:
        ctx.solve(b)
    finally:
        ctx.set_schur_precision(0)
    ctx.solve(c2)
    np.testing.assert_array_equal(a.frame_state, c2.frame_state)
    np.testing.assert_array_equal(a.inv_depth, c2.inv_depth)
    assert not np.array_equal(a.frame_state, b.frame_state)
    np.testing.assert_allclose(b.frame_state, a.frame_state, rtol=1e-5, atol=1e-7)
    with pytest.raises(Exception):
        ctx.set_schur_precision(7)


def test_overlapped_solve_runs_the_host_work_once_and_changes_nothing(ctx):
    """xrhip_ba_solve_overlapped: the caller's host work runs exactly once beside the device's (the sliding-window tracker puts the
    next detection's corner selection there), for the single-launch solves, the multi-launch window solves and a problem that needs
    no launch at all; states and summaries are those of xrhip_ba_solve, bit for bit."""
    loc = bs.make_localize(seed=9)[0]
    win = bs.make_window(K=8, L=100, seed=43)[0]
    fixed = bs.make_localize(seed=9)[0]
    fixed.frame_fix[:] = 3                      # nothing free: the solve returns without a launch
    for pd in (loc, win, fixed):
        a, b = pd.copy(), pd.copy()
        sa = ctx.solve(a)
        calls = []
        sb = ctx.solve(b, host_work=lambda: calls.append(1))
        assert calls == [1]
        np.testing.assert_array_equal(a.frame_state, b.frame_state)
        np.testing.assert_array_equal(a.inv_depth, b.inv_depth)
        assert (sa.iterations, sa.termination, sa.usable) == (sb.iterations, sb.termination, sb.usable)
        assert sa.final_cost == sb.final_cost


def test_chained_solves_equal_one_after_the_other(ctx):
    """xrhip_ba_solve_chained (localize_newframe + refine_subwindow as one submission): the second problem's linked frame starts from
    the first solve's result ON THE DEVICE.  Same bits as solving the first, copying the frame's state over on the host and solving
    the second -- for the pipeline's own frozen pair of problems, with the host work run once, with an integration queued behind the
    second solve, and through the general form when a problem is not a single-launch solve."""
    from tests import ba_snapshots
    from xrslam_amd import ba as _ba
    snaps = {name: pd for name, pd, _ in ba_snapshots.load_all()}
    loc, sub, win = snaps["s1_localize"], snaps["s1_subwindow"], snaps["s1_window"]
    ctx2 = _ba.BaContext(max_frames=32, max_landmarks=2048, max_obs=16384)
    _, truth = bs.make_window(K=5, L=20, seed=31)
    smp, t_end = truth["samples"][1], truth["times"][2]
    try:
        free_loc = [f for f in range(len(loc.frame_state)) if loc.frame_fix[f] != 3][-1]
        free_sub = [f for f in range(len(sub.frame_state)) if sub.frame_fix[f] != 3][-1]
        for first, lf, second, ls in ((loc, free_loc, sub, free_sub), (loc, free_loc, win, len(win.frame_state) - 1),
                                      (sub, free_sub, loc, free_loc)):
            a1, a2 = first.copy(), second.copy()
            m1 = ctx.solve(a1)
            a2.frame_state[ls] = a1.frame_state[lf]
            m2 = ctx.solve(a2)
            b1, b2 = first.copy(), second.copy()
            b2.frame_state[ls] = 123.0                      # whatever the linked frame holds on entry is ignored
            calls = []
            n1, n2 = ctx2.solve_chained(b1, lf, ctx, b2, ls, host_work=lambda: calls.append(1))
            assert calls == [1]
            np.testing.assert_array_equal(b1.frame_state, a1.frame_state)
            np.testing.assert_array_equal(b2.frame_state, a2.frame_state)
            np.testing.assert_array_equal(b2.inv_depth, a2.inv_depth)
            assert (m1.iterations, m1.termination, m1.final_cost) == (n1.iterations, n1.termination, n1.final_cost)
            assert (m2.iterations, m2.termination, m2.final_cost) == (n2.iterations, n2.termination, n2.final_cost)
            assert not np.array_equal(a2.frame_state[ls], second.frame_state[ls])     # and the hand-over mattered
        # an integration queued behind the pair starts from the SECOND solve's biases
        a1, a2 = loc.copy(), sub.copy()
        ctx.solve(a1)
        a2.frame_state[free_sub] = a1.frame_state[free_loc]
        ctx.solve(a2)
        direct = ctx.preintegrate(smp, t_end, a2.frame_state[free_sub, 10:13], a2.frame_state[free_sub, 13:16], bs.NOISE36)
        b1, b2 = loc.copy(), sub.copy()
        ctx.preintegrate_after_solve(smp, t_end, free_sub, bs.NOISE36)
        ctx2.solve_chained(b1, free_loc, ctx, b2, free_sub)
        np.testing.assert_array_equal(ctx.preintegrate_end(), direct)
        np.testing.assert_array_equal(b2.frame_state, a2.frame_state)
        # begin / end on their own: a single-launch problem is queued and collected with xrhip_ba_solve's result; a window problem is
        # not begun; a begun solve blocks the context until it is collected or aborted
        a, b = loc.copy(), loc.copy()
        ma = ctx.solve(a)
        assert ctx2.solve_begin(b) is True
        with pytest.raises(Exception):
            ctx2.solve(loc.copy())
        mb = ctx2.solve_end()
        np.testing.assert_array_equal(a.frame_state, b.frame_state)
        assert (ma.iterations, ma.final_cost) == (mb.iterations, mb.final_cost)
        assert ctx2.solve_begin(win.copy()) is False
        with pytest.raises(Exception):
            ctx2.solve_end()                                  # nothing in flight
        c = loc.copy()
        assert ctx2.solve_begin(c) is True
        ctx2.solve_abort()
        np.testing.assert_array_equal(c.frame_state, loc.frame_state)      # an aborted solve writes nothing back
        ctx2.solve(c)
        np.testing.assert_array_equal(c.frame_state, a.frame_state)
        with pytest.raises(Exception):
            ctx.solve_chained(loc.copy(), free_loc, ctx, sub.copy(), free_sub)      # two distinct contexts are needed
        with pytest.raises(Exception):
            ctx2.solve_chained(loc.copy(), len(loc.frame_state), ctx, sub.copy(), free_sub)
    finally:
        ctx2.close()


def test_early_delta_of_a_queued_integration(ctx):
    """xrhip_ba_preintegrate_early: the delta (dt, dq, dp, dv) of a queued integration is available before the record's covariance
    part and equals -- bit for bit -- both the finished record's and the Jacobian-free integration of the same samples (what
    FeatureTracker::work and mirror_frame used to compute in two launches)."""
    _, truth = bs.make_window(K=5, L=20, seed=31)
    for k in (1, 2, 3):
        smp, t_end = truth["samples"][k], truth["times"][k + 1]
        bg, ba_ = np.array([1e-3, -2e-3, 5e-4]), np.array([0.02, -0.01, 0.03])
        plain = ctx.preintegrate(smp, t_end, bg, ba_, bs.NOISE36, jac=False, cov=False)
        ctx.preintegrate_begin(smp, t_end, bg, ba_, bs.NOISE36)
        early = ctx.preintegrate_early(0)
        full = ctx.preintegrate_end()
        np.testing.assert_array_equal(early, full[:11])
        np.testing.assert_array_equal(early, plain[:11])
        assert np.abs(full[56:]).max() > 0
    with pytest.raises(Exception):
        ctx.preintegrate_early(0)                  # nothing in flight


def test_one_preintegration_batch_in_flight_per_context(ctx):
    """A second xrhip_ba_preintegrate_begin without _end is refused (it would overwrite the staging block the first batch's kernel
    writes its record into); xrhip_ba_preintegrate_cancel releases the context; a fresh begin / end then gives the blocking call's
    record."""
    import ctypes as C
    from xrslam_amd import ba as _ba
    rng = np.random.RandomState(5)
    n = 10
    smp = np.zeros((n, 7))
    smp[:, 0] = 1.0 + 0.005 * np.arange(n)
    smp[:, 1:4] = 0.1 * rng.randn(n, 3)
    smp[:, 4:7] = np.array([0.0, 0.0, 9.8]) + 0.2 * rng.randn(n, 3)
    t_end = float(smp[-1, 0] + 0.005)
    ref = ctx.preintegrate(smp, t_end, np.zeros(3), np.zeros(3), bs.NOISE36)
    lib = _ba.L()
    vp = C.c_void_p
    begin = np.zeros(1, np.int32)
    count = np.array([n], np.int32)
    te = np.array([t_end])
    z3 = np.zeros(3)
    s = np.ascontiguousarray(smp)
    noise = np.ascontiguousarray(bs.NOISE36, np.float64)

    def p(a):
        return a.ctypes.data_as(vp)
    args = (ctx._h, p(s), p(begin), p(count), p(te), p(z3), p(z3), 1, p(noise), 1, 1)
    assert lib.xrhip_ba_preintegrate_begin(*args) == 0
    assert lib.xrhip_ba_preintegrate_begin(*args) != 0          # XRHIP_ESTATE: one batch in flight
    assert lib.xrhip_ba_preintegrate_cancel(ctx._h) == 0
    assert lib.xrhip_ba_preintegrate_begin(*args) == 0
    out = np.zeros(ref.size)
    assert lib.xrhip_ba_preintegrate_end(ctx._h, p(out)) == 0
    np.testing.assert_array_equal(out.reshape(ref.shape), ref)
