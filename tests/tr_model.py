"""A second, independent implementation of the minimiser behind `Solver::solve`.

oracle/ba_oracle.cpp restates Ceres 1.14's trust-region loop in C++ and the HIP kernels follow it; both come from the same
reading of Ceres, so agreeing with each other proves nothing about Ceres.  This module is that loop written a second time, from the
Ceres documentation (Solver::Options, "Trust Region Methods", "Dogleg", LossFunction / Corrector, IterationSummary) and the
structure of its `TrustRegionMinimizer` / `DoglegStrategy` -- NOT from ba_oracle.cpp, and with different machinery:

  * dense numpy normal equations over every free block (no Schur complement, no elimination ordering, numpy's Cholesky),
  * its own block bookkeeping, Jacobi scaling, Cauchy-loss corrector, dogleg (traditional) step with the mu-regularised
    Gauss-Newton solve, step evaluation, radius update and the three convergence tests,
  * its own marginalisation-prior factor (reference ceres/marginalization_factor.h:27-72) and quaternion algebra.

What it shares with the oracle are the per-factor residual / Jacobian evaluators of the three measurement factors
(oracle.ba_oracle.eval_reprojection / eval_rotation / eval_imu -- restatements of reference ceres/reprojection_factor.h,
rotation_factor.h, preintegration_factor.h that tests/test_oracle_ba.py checks against finite differences).

The reference's one behavioural quirk is modelled explicitly: parameter blocks ARE the frames' members
(estimation/solver.cpp:84-106), the IMU factor takes its bias linearisation point from `frame_i->motion.bg / ba`
(ceres/preintegration_factor.h:38-40) and `update_state_every_iteration = true` (solver.cpp:187) makes Ceres copy the iterate
into those members after every SUCCESSFUL iteration -- i.e. after the new point has already been linearised with the OLD reference.

tests/test_tr_model.py replays the frozen pipeline problems (tests/golden/ba_snapshots) through this model and requires the
committed per-iteration record of the oracle: costs, model cost change, relative decrease, radius, accept / reject, termination.
TEST INFRASTRUCTURE ONLY."""
import numpy as np

from oracle import ba_oracle as bo

# ceres::Solver::Options defaults the reference leaves untouched (docs: "Solver::Options")
INITIAL_RADIUS, MAX_RADIUS, MIN_RADIUS = 1e4, 1e16, 1e-32
MIN_RELATIVE_DECREASE = 1e-3
FUNCTION_TOLERANCE, GRADIENT_TOLERANCE, PARAMETER_TOLERANCE = 1e-6, 1e-10, 1e-8
MAX_CONSECUTIVE_INVALID_STEPS = 5
# DoglegStrategy
MIN_DIAGONAL, MAX_DIAGONAL = 1e-6, 1e32
MIN_MU, MAX_MU, MU_INCREASE = 1e-8, 1.0, 10.0
DECREASE_THRESHOLD, INCREASE_THRESHOLD = 0.25, 0.75

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2


# ------------------------------------------------------------------------------------------------ quaternions (x, y, z, w)
def q_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def q_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def expmap(w):
    a = np.linalg.norm(w)
    if a == 0.0:
        return np.array([0.0, 0.0, 0.0, 1.0])
    return np.concatenate([np.sin(0.5 * a) * w / a, [np.cos(0.5 * a)]])


def logmap(q):
    v = q[:3]
    n = np.linalg.norm(v)
    if n == 0.0:
        return np.zeros(3)
    angle = 2.0 * np.arctan2(n, abs(q[3]))      # Eigen::AngleAxisd(q): angle in [0, pi], axis flipped when w < 0
    return (v / (n if q[3] >= 0 else -n)) * angle


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def right_jacobian(w):
    a = np.linalg.norm(w)
    if a < 1e-5:
        return np.eye(3) - 0.5 * hat(w) + hat(w) @ hat(w) / 6.0
    h = hat(w)
    return np.eye(3) - (1 - np.cos(a)) / (a * a) * h + (a - np.sin(a)) / (a ** 3) * (h @ h)


def plus(state, d15, pose_free, motion_free):
    """QuaternionParameterization::Plus (q <- (q expmap(dq)).normalized()) and the additive blocks."""
    s = np.array(state, float)
    if pose_free:
        q = q_mul(s[:4], expmap(d15[:3]))
        s[:4] = q / np.linalg.norm(q)
        s[4:7] += d15[3:6]
    if motion_free:
        s[7:16] += d15[6:15]
    return s


# ------------------------------------------------------------------------------------------------ the problem
class Program:
    """Ceres' reduced program: the non-constant parameter blocks that some residual block uses, in (frame pose, frame motion,
    landmark) order -- the order only permutes the normal equations."""

    def __init__(self, pd):
        self.pd = pd
        F, L = len(pd.frame_state), len(pd.inv_depth)
        self.pose_free = np.array([(int(x) & 1) == 0 for x in pd.frame_fix])
        self.motion_free = np.array([(int(x) & 2) == 0 for x in pd.frame_fix])
        self.lm_free = np.array([int(x) == 0 for x in pd.landmark_fix]) if L else np.zeros(0, bool)
        pose_used, motion_used, lm_used = np.zeros(F, bool), np.zeros(F, bool), np.zeros(max(L, 1), bool)
        for o in range(len(pd.obs_tgt)):
            pose_used[pd.obs_tgt[o]] = pose_used[pd.obs_ref[o]] = True
            lm_used[pd.obs_lm[o]] = True
        for o in range(len(pd.rot_tgt)):
            pose_used[pd.rot_tgt[o]] = True
        for k in range(len(pd.imu_i)):
            for f in (pd.imu_i[k], pd.imu_j[k]):
                pose_used[f] = motion_used[f] = True
        for f in pd.prior_frames:
            pose_used[f] = motion_used[f] = True
        self.pose_off, self.motion_off, self.lm_off = -np.ones(F, int), -np.ones(F, int), -np.ones(max(L, 1), int)
        n = 0
        for f in range(F):
            if self.pose_free[f] and pose_used[f]:
                self.pose_off[f] = n
                n += 6
            if self.motion_free[f] and motion_used[f]:
                self.motion_off[f] = n
                n += 9
        for l in range(L):
            if self.lm_free[l] and lm_used[l]:
                self.lm_off[l] = n
                n += 1
        self.n = n

    # -- user state <-> iterate
    def apply(self, states, depths, step):
        """Plus(x, step) on every free block."""
        pd, S, D = self.pd, np.array(states), np.array(depths)
        for f in range(len(S)):
            d15 = np.zeros(15)
            pf, mf = self.pose_off[f] >= 0, self.motion_off[f] >= 0
            if pf:
                d15[:6] = step[self.pose_off[f]:self.pose_off[f] + 6]
            if mf:
                d15[6:] = step[self.motion_off[f]:self.motion_off[f] + 9]
            if pf or mf:
                S[f] = plus(S[f], d15, pf, mf)
        for l in range(len(D)):
            if self.lm_off[l] >= 0:
                D[l] += step[self.lm_off[l]]
        return S, D

    def ambient(self, states, depths):
        """the free blocks in their ambient (global) coordinates, concatenated: what Ceres' x vector holds"""
        out = []
        for f in range(len(states)):
            if self.pose_off[f] >= 0:
                out.append(states[f][:7])
            if self.motion_off[f] >= 0:
                out.append(states[f][7:16])
        for l in range(len(depths)):
            if self.lm_off[l] >= 0:
                out.append([depths[l]])
        return np.concatenate(out) if out else np.zeros(0)

    # -- evaluation: cost, and (optionally) gradient g = J^T r and H = J^T J of the CORRECTED residual blocks
    def evaluate(self, states, depths, bias_ref, want_jac):
        pd, n = self.pd, self.n
        H = np.zeros((n, n)) if want_jac else None
        g = np.zeros(n) if want_jac else None
        cost = 0.0

        def add(blocks, r):
            # blocks: [(offset, J)] of the free parameter blocks of one residual block
            for oa, Ja in blocks:
                g[oa:oa + Ja.shape[1]] += Ja.T @ r
                for ob, Jb in blocks:
                    H[oa:oa + Ja.shape[1], ob:ob + Jb.shape[1]] += Ja.T @ Jb

        cam, sic = pd.cam_ext, pd.sqrt_inv_cov
        for o in range(len(pd.obs_tgt)):
            ft, fr, l = pd.obs_tgt[o], pd.obs_ref[o], pd.obs_lm[o]
            ot, orf, ol = self.pose_off[ft], self.pose_off[fr], self.lm_off[l]
            if ot < 0 and orf < 0 and ol < 0:
                continue                                   # every block constant: removed by the preprocessor
            r, Jt, Jr, Jl = bo.eval_reprojection(states[ft], states[fr], depths[l], pd.obs_z_tgt[o], pd.obs_z_ref[o], cam, sic,
                                                 jac=want_jac)
            s = float(r @ r)
            cost += 0.5 * np.log1p(s)                      # CauchyLoss(1): rho(s) = log(1 + s)
            if want_jac:
                w = np.sqrt(1.0 / (1.0 + s))               # Corrector: rho'' < 0 => residual and Jacobian scaled by sqrt(rho')
                blocks = []
                if ot >= 0:
                    blocks.append((ot, w * Jt))
                if orf >= 0 and fr != ft:
                    blocks.append((orf, w * Jr))
                if ol >= 0:
                    blocks.append((ol, w * Jl.reshape(2, 1)))
                add(blocks, w * r)
        for o in range(len(pd.rot_tgt)):
            ft, fr = pd.rot_tgt[o], pd.rot_ref[o]
            ot = self.pose_off[ft]
            if ot < 0:
                continue
            r, Jq = bo.eval_rotation(states[ft], states[fr], pd.rot_z_tgt[o], pd.rot_z_ref[o], cam, sic, jac=want_jac)
            s = float(r @ r)
            cost += 0.5 * np.log1p(s)
            if want_jac:
                w = np.sqrt(1.0 / (1.0 + s))
                J6 = np.zeros((2, 6))
                J6[:, :3] = w * Jq
                add([(ot, J6)], w * r)
        for k in range(len(pd.imu_i)):
            fi, fj = pd.imu_i[k], pd.imu_j[k]
            offs = [(self.pose_off[fi], 0, 6, 0), (self.motion_off[fi], 6, 15, 0), (self.pose_off[fj], 0, 6, 1),
                    (self.motion_off[fj], 6, 15, 1)]
            if all(o < 0 for o, _, _, _ in offs):
                continue
            r, Ji, Jj = bo.eval_imu(states[fi], states[fj], pd.imu_data[k], bias_ref[k], pd.imu_ext, jac=want_jac)
            cost += 0.5 * float(r @ r)                     # no loss function on IMU factors (solver.cpp:142-157)
            if want_jac:
                add([(o, (Ji, Jj)[side][:, a:b]) for o, a, b, side in offs if o >= 0], r)
        NP = len(pd.prior_frames)
        if NP:
            delta = np.zeros(15 * NP)
            Jloc = np.zeros((15 * NP, 15 * NP))
            for i, f in enumerate(pd.prior_frames):
                lin, st = pd.prior_lin[i], states[f]
                rq = logmap(q_mul(q_conj(lin[:4]), st[:4]))
                delta[15 * i:15 * i + 3] = rq
                delta[15 * i + 3:15 * i + 15] = st[4:16] - lin[4:16]
                Jloc[15 * i:15 * i + 3, 15 * i:15 * i + 3] = np.linalg.inv(right_jacobian(rq))
                Jloc[15 * i + 3:15 * i + 15, 15 * i + 3:15 * i + 15] = np.eye(12)
            r = pd.prior_sqrt_info @ delta + pd.prior_infovec
            cost += 0.5 * float(r @ r)
            if want_jac:
                J = pd.prior_sqrt_info @ Jloc
                blocks = []
                for i, f in enumerate(pd.prior_frames):
                    if self.pose_off[f] >= 0:
                        blocks.append((self.pose_off[f], J[:, 15 * i:15 * i + 6]))
                    if self.motion_off[f] >= 0:
                        blocks.append((self.motion_off[f], J[:, 15 * i + 6:15 * i + 15]))
                add(blocks, r)
        return cost, g, H


def solve(pd, max_iterations=None, trace=None, refresh_bias_reference=True):
    """Runs the minimiser on an abi.BaProblemData IN PLACE (like the reference); returns a dict with iterations,
    successful_steps, termination, initial_cost, final_cost.  trace: list receiving, per trial that reached the accept / reject
    decision, (iteration, x_cost, candidate_cost, model_cost_change, relative_decrease, radius, step_norm, mu, accepted).
    refresh_bias_reference=False: the IMU factors keep the bias reference of the solve's start (NOT the reference's behaviour;
    tests use it to show what the refresh does)."""
    prog = Program(pd)
    max_it = int(pd.max_iterations if max_iterations is None else max_iterations)
    x_states, x_depths = np.array(pd.frame_state), np.array(pd.inv_depth)
    out = dict(iterations=0, successful_steps=0, termination=CONVERGENCE)
    if prog.n == 0:
        out.update(initial_cost=0.0, final_cost=0.0)
        return out
    user_bias = lambda S: np.array([S[pd.imu_i[k]][10:16] for k in range(len(pd.imu_i))]).reshape(-1, 6)   # noqa: E731
    bias_ref = user_bias(x_states)                          # the frames' members as the solve starts

    # ---- IterationZero: cost, gradient, Jacobian; Jacobi scaling from THIS Jacobian, kept for the whole solve
    x_cost, g, H = prog.evaluate(x_states, x_depths, bias_ref, True)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    out["initial_cost"] = x_cost

    def gradient_max_norm(S, D, grad):
        S2, D2 = prog.apply(S, D, -grad)
        return float(np.max(np.abs(prog.ambient(S, D) - prog.ambient(S2, D2))))

    gmax = gradient_max_norm(x_states, x_depths, g)
    x_norm = float(np.linalg.norm(prog.ambient(x_states, x_depths)))
    radius, mu, reuse = INITIAL_RADIUS, MIN_MU, False
    iteration, invalid, last_successful = 0, 0, True        # iteration 0 counts as successful for the gradient test
    dl = {}                                                  # what DoglegStrategy keeps between calls while reuse is set

    def finish(term):
        pd.frame_state[:] = x_states
        pd.inv_depth[:] = x_depths
        out.update(iterations=iteration, termination=term, final_cost=x_cost)
        return out

    if gmax <= GRADIENT_TOLERANCE:
        return finish(CONVERGENCE)
    while True:
        # ---- FinalizeIterationAndCheckIfMinimizerCanContinue (callbacks first: the state-updating one refreshes the user state
        # -- hence the IMU factors' bias reference -- after a successful iteration)
        if last_successful and refresh_bias_reference:
            bias_ref = user_bias(x_states)
        if iteration >= max_it:
            return finish(NO_CONVERGENCE)
        if last_successful and gmax <= GRADIENT_TOLERANCE:
            return finish(CONVERGENCE)
        if radius < MIN_RADIUS:
            return finish(CONVERGENCE)
        iteration += 1
        # ---- DoglegStrategy::ComputeStep on the column-scaled Jacobian: Hs = S H S, gs = S g
        if not reuse:
            Hs, gs = H * np.outer(scale, scale), g * scale
            diag = np.sqrt(np.clip(np.diag(Hs), MIN_DIAGONAL, MAX_DIAGONAL))
            grad = gs / diag                                # gradient in the D-scaled space
            Jg2 = float((grad / diag) @ Hs @ (grad / diag))  # |J D^-1 grad|^2
            alpha = float(grad @ grad) / Jg2
            gn, ok = None, False
            while mu < MAX_MU:
                try:
                    Lc = np.linalg.cholesky(Hs + mu * np.diag(diag * diag))
                    sol = np.linalg.solve(Lc.T, np.linalg.solve(Lc, gs))
                    if np.all(np.isfinite(sol)):
                        gn, ok = -sol * diag, True           # scaled Gauss-Newton step D * (-x)
                        break
                except np.linalg.LinAlgError:
                    pass
                mu *= MU_INCREASE
            dl = dict(Hs=Hs, gs=gs, diag=diag, grad=grad, alpha=alpha, gn=gn, ok=ok)
            reuse = True
        step = None
        if dl["ok"]:
            grad, gn, diag, alpha = dl["grad"], dl["gn"], dl["diag"], dl["alpha"]
            gnorm, gn_norm = float(np.linalg.norm(grad)), float(np.linalg.norm(gn))
            if gn_norm <= radius:
                d_step, d_norm = gn.copy(), gn_norm
            elif gnorm * alpha >= radius:
                d_step, d_norm = -(radius / gnorm) * grad, radius
            else:
                b_dot_a = -alpha * float(grad @ gn)
                a2 = (alpha * gnorm) ** 2
                bma2 = a2 - 2.0 * b_dot_a + gn_norm ** 2
                c = b_dot_a - a2
                d = np.sqrt(c * c + bma2 * (radius ** 2 - a2))
                beta = (d - c) / bma2 if c <= 0 else (radius ** 2 - a2) / (d + c)
                d_step = (-alpha * (1.0 - beta)) * grad + beta * gn
                d_norm = float(np.linalg.norm(d_step))
            step = d_step / diag
        # ---- model cost change of the step: -(J step)^T (r + J step / 2)
        valid = False
        if step is not None:
            model_change = -(float(step @ dl["gs"]) + 0.5 * float(step @ dl["Hs"] @ step))
            valid = model_change > 0.0
        if not valid:
            invalid += 1
            if invalid >= MAX_CONSECUTIVE_INVALID_STEPS:
                return finish(FAILURE)
            mu *= MU_INCREASE                               # DoglegStrategy::StepIsInvalid
            reuse = False
            last_successful = False
            continue
        invalid = 0
        delta = step * scale                                # back to the unscaled local coordinates
        c_states, c_depths = prog.apply(x_states, x_depths, delta)
        cand_cost, _, _ = prog.evaluate(c_states, c_depths, bias_ref, False)
        # ---- ParameterToleranceReached / FunctionToleranceReached
        step_norm = float(np.linalg.norm(prog.ambient(x_states, x_depths) - prog.ambient(c_states, c_depths)))
        if step_norm <= PARAMETER_TOLERANCE * (x_norm + PARAMETER_TOLERANCE):
            return finish(CONVERGENCE)
        if abs(x_cost - cand_cost) <= FUNCTION_TOLERANCE * x_cost:
            return finish(CONVERGENCE)
        rel = (x_cost - cand_cost) / model_change
        accepted = rel > MIN_RELATIVE_DECREASE
        if trace is not None:
            trace.append((iteration, x_cost, cand_cost, model_change, rel, radius, step_norm, mu, 1.0 if accepted else 0.0))
        if accepted:
            x_states, x_depths, x_cost = c_states, c_depths, cand_cost
            x_norm = float(np.linalg.norm(prog.ambient(x_states, x_depths)))
            # EvaluateGradientAndJacobian -- the user state (bias reference) is still the previous iterate's here
            x_cost_lin, g, H = prog.evaluate(x_states, x_depths, bias_ref, True)
            x_cost = x_cost_lin
            gmax = gradient_max_norm(x_states, x_depths, g)
            out["successful_steps"] += 1
            # DoglegStrategy::StepAccepted
            if rel < DECREASE_THRESHOLD:
                radius *= 0.5
            if rel > INCREASE_THRESHOLD:
                radius = max(radius, 3.0 * d_norm)
            radius = min(MAX_RADIUS, radius)
            mu = max(MIN_MU, 2.0 * mu / MU_INCREASE)
            reuse = False
            last_successful = True
        else:
            radius *= 0.5                                   # DoglegStrategy::StepRejected
            reuse = True
            last_successful = False
