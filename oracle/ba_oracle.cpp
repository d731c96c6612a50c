// ba_oracle.cpp -- CPU restatement of the reference's sliding-window VI bundle adjustment.
//
// TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg as the checker of the HIP path.  Never linked into the product.
//
// Reference-owned math restated line by line (double precision, single thread):
//   PreIntegrator::increment/integrate/compute_sqrt_inv_cov   estimation/preintegrator.cpp:22-100
//   CeresReprojectionErrorFactor / ...PriorFactor::Evaluate    estimation/ceres/reprojection_factor.h:25-123
//   CeresRotationPriorFactor::Evaluate                         estimation/ceres/rotation_factor.h:23-59
//   CeresPreIntegrationErrorFactor / ...PriorFactor::Evaluate  estimation/ceres/preintegration_factor.h:20-199
//   CeresMarginalizationFactor::Evaluate / marginalize         estimation/ceres/marginalization_factor.h:27-475
//   QuaternionParameterization::Plus                           estimation/ceres/quaternion_parameterization.h:11-17
//   Solver::solve options                                      estimation/solver.cpp:176-190
//
// Upstream semantics restated from the published algorithm (Ceres-solver 1.14.0 @ e809cf0,
// cmake/depends/ceres-solver.cmake:7-8, NOT under /root/reference; written from its public
// sources' documented behaviour -- PARITY UNPINNED):
//   trust_region_minimizer.cc (iteration structure, tolerances, step acceptance),
//   dogleg_strategy.cc (traditional dogleg, mu regularisation, radius update),
//   corrector.cc + loss_function.cc (CauchyLoss(1): rho'' < 0 => sqrt(rho') scaling),
//   schur_eliminator (landmark e-blocks eliminated, dense reduced system),
//   trust_region_preprocessor.cc (constant blocks removed), callbacks.cc
//   (StateUpdatingCallback: user state refreshed after every successful iteration, which
//   moves the bias reference read by the IMU factor -- see bias_ref below).
//   Defaults: initial radius 1e4, max radius 1e16, min_relative_decrease 1e-3,
//   function_tolerance 1e-6, gradient_tolerance 1e-10, parameter_tolerance 1e-8,
//   jacobi_scaling, monotonic steps, max_num_consecutive_invalid_steps 5,
//   dogleg min_diagonal 1e-6, max_diagonal 1e32, mu in [1e-8, 1] x10.
//
// The reference's own tests pin none of this: the oracle is validated
// by finite-difference Jacobian checks, closed-form pre-integration cases and
// marginalise/solve commutation (tests/test_oracle_ba.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "../include/xrslam_hip.h"
#include "la.hpp"

using namespace orc;

namespace {

const double GRAVITY_NOMINAL = 9.80665;   // common.h:41

enum { ES_Q = 0, ES_P = 3, ES_V = 6, ES_BG = 9, ES_BA = 12, ES_SIZE = 15 };

struct FrameState {
    Quat q;
    Vec3 p, v, bg, ba;
};

FrameState load_state(const double *s) {
    FrameState f;
    f.q = Quat::from_xyzw(s);
    f.p = vec3(s[4], s[5], s[6]);
    f.v = vec3(s[7], s[8], s[9]);
    f.bg = vec3(s[10], s[11], s[12]);
    f.ba = vec3(s[13], s[14], s[15]);
    return f;
}
void store_state(const FrameState &f, double *s) {
    f.q.to_xyzw(s);
    for (int i = 0; i < 3; ++i) {
        s[4 + i] = f.p[i];
        s[7 + i] = f.v[i];
        s[10 + i] = f.bg[i];
        s[13 + i] = f.ba[i];
    }
}

struct Extrinsic {
    Quat q;
    Vec3 p;
};

struct ImuFactorData {
    double dt;
    Quat dq;
    Vec3 dp, dv;
    Mat3 dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba;
    Mat<15, 15> sqrt_inv_cov;
};

ImuFactorData load_imu(const double *d) {
    ImuFactorData f;
    f.dt = d[0];
    f.dq = Quat::from_xyzw(d + 1);
    f.dp = vec3(d[5], d[6], d[7]);
    f.dv = vec3(d[8], d[9], d[10]);
    const double *j = d + 11;
    Mat3 *ms[5] = {&f.dq_dbg, &f.dp_dbg, &f.dp_dba, &f.dv_dbg, &f.dv_dba};
    for (int k = 0; k < 5; ++k)
        for (int i = 0; i < 9; ++i) ms[k]->a[i] = j[9 * k + i];
    for (int i = 0; i < 225; ++i) f.sqrt_inv_cov.a[i] = d[56 + i];
    return f;
}

// ------------------------------------------------------------ pre-integration
struct PreInt {
    Mat3 cov_w, cov_a, cov_bg, cov_ba;
    double t = 0;
    Quat q;
    Vec3 p, v;
    Mat<15, 15> cov, sqrt_inv_cov;
    Mat3 dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba;

    void increment(double dt, const Vec3 &w_raw, const Vec3 &a_raw, const Vec3 &bg, const Vec3 &ba, bool jac,
                   bool cv) {   // preintegrator.cpp:22-76
        Vec3 w = w_raw - bg;
        Vec3 a = a_raw - ba;
        Mat3 dR = q.matrix();
        if (cv) {
            Mat<9, 9> A = Mat<9, 9>::identity();
            A.set_block<3, 3>(ES_Q, ES_Q, expmap(w * dt).conjugate().matrix());
            A.set_block<3, 3>(ES_V, ES_Q, (dR * hat(a)) * (-dt));
            A.set_block<3, 3>(ES_P, ES_Q, (dR * hat(a)) * (-0.5 * dt * dt));
            A.set_block<3, 3>(ES_P, ES_V, Mat3::identity() * dt);
            Mat<9, 6> B;
            B.set_block<3, 3>(ES_Q, 0, right_jacobian(w * dt) * dt);
            B.set_block<3, 3>(ES_V, 3, dR * dt);
            B.set_block<3, 3>(ES_P, 3, dR * (0.5 * dt * dt));
            Mat<6, 6> wn;
            double inv_dt = 1.0 / std::max(dt, 1.0e-7);
            wn.set_block<3, 3>(0, 0, cov_w * inv_dt);
            wn.set_block<3, 3>(3, 3, cov_a * inv_dt);
            Mat<9, 9> c9 = cov.block<9, 9>(0, 0);
            Mat<9, 9> n9 = A * c9 * A.t() + B * wn * B.t();
            cov.set_block<9, 9>(0, 0, n9);
            cov.add_block<3, 3>(ES_BG, ES_BG, cov_bg * dt);
            cov.add_block<3, 3>(ES_BA, ES_BA, cov_ba * dt);
        }
        if (jac) {
            dp_dbg += dv_dbg * dt - (dR * hat(a) * dq_dbg) * (0.5 * dt * dt);
            dp_dba += dv_dba * dt - dR * (0.5 * dt * dt);
            dv_dbg -= (dR * hat(a) * dq_dbg) * dt;
            dv_dba -= dR * dt;
            dq_dbg = expmap(w * dt).conjugate().matrix() * dq_dbg - right_jacobian(w * dt) * dt;
        }
        t = t + dt;
        p = p + v * dt + (q * a) * (0.5 * dt * dt);
        v = v + (q * a) * dt;
        q = (q * expmap(w * dt)).normalized();
    }
};

// -------------------------------------------------------- reprojection factor
// CeresReprojectionErrorFactor::Evaluate.  J blocks: tgt (2x6: q local 3, p 3), ref (2x6), depth (2x1).
void eval_reprojection(const FrameState &tgt, const FrameState &ref, double inv_depth, const Vec3 &z_tgt,
                       const Vec3 &z_ref, const Extrinsic &cam, const double sic[2], double r[2], double *Jt,
                       double *Jr, double *Jl) {
    Vec3 b1, b2;
    s2_tangential_basis(z_tgt, b1, b2);
    Mat3 T;   // local_tangent = [b1 b2 z]
    for (int i = 0; i < 3; ++i) {
        T(i, 0) = b1[i];
        T(i, 1) = b2[i];
        T(i, 2) = z_tgt[i];
    }
    Vec3 y_ref = z_ref / inv_depth;
    Vec3 y_ref_center = cam.q * y_ref + cam.p;
    Vec3 x = ref.q * y_ref_center + ref.p;
    Vec3 y_tgt_center = tgt.q.conjugate() * (x - tgt.p);
    Vec3 y_tgt = cam.q.conjugate() * (y_tgt_center - cam.p);
    Vec3 u = T.t() * y_tgt;
    double rx = u[0] / u[2], ry = u[1] / u[2];
    if (Jt || Jr || Jl) {
        Mat<2, 3> dproj;
        dproj(0, 0) = 1.0 / u[2];
        dproj(0, 2) = -u[0] / (u[2] * u[2]);
        dproj(1, 1) = 1.0 / u[2];
        dproj(1, 2) = -u[1] / (u[2] * u[2]);
        Mat<2, 2> S;
        S(0, 0) = sic[0];
        S(1, 1) = sic[1];
        Mat<2, 3> dr_dy_tgt = S * dproj * T.t();
        Mat<2, 3> dr_dy_tgt_center = dr_dy_tgt * cam.q.conjugate().matrix();
        Mat<2, 3> dr_dx = dr_dy_tgt_center * tgt.q.conjugate().matrix();
        Mat<2, 3> dr_dy_ref_center = dr_dx * ref.q.matrix();
        if (Jt) {
            Mat<2, 3> a = dr_dy_tgt_center * hat(y_tgt_center);
            Mat<2, 3> b = -dr_dx;
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    Jt[i * 6 + j] = a(i, j);
                    Jt[i * 6 + 3 + j] = b(i, j);
                }
        }
        if (Jr) {
            Mat<2, 3> a = -(dr_dy_ref_center * hat(y_ref_center));
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    Jr[i * 6 + j] = a(i, j);
                    Jr[i * 6 + 3 + j] = dr_dx(i, j);
                }
        }
        if (Jl) {
            Mat<2, 1> a = -(dr_dy_ref_center * cam.q.matrix() * y_ref) / inv_depth;
            Jl[0] = a[0];
            Jl[1] = a[1];
        }
    }
    r[0] = sic[0] * rx;
    r[1] = sic[1] * ry;
}

// CeresRotationPriorFactor::Evaluate.  J: 2x3 (q_tgt local)
void eval_rotation(const FrameState &tgt, const FrameState &ref, const Vec3 &z_tgt, const Vec3 &z_ref,
                   const Extrinsic &cam, const double sic[2], double r[2], double *Jq) {
    Vec3 b1, b2;
    s2_tangential_basis(z_tgt, b1, b2);
    Mat3 T;
    for (int i = 0; i < 3; ++i) {
        T(i, 0) = b1[i];
        T(i, 1) = b2[i];
        T(i, 2) = z_tgt[i];
    }
    Vec3 z_ref_center = cam.q * z_ref + cam.p;
    Vec3 z_tgt_center = (tgt.q.conjugate() * ref.q) * z_ref_center;
    Vec3 z_t = cam.q.conjugate() * (z_tgt_center - cam.p);
    Vec3 u = T.t() * z_t;
    double rx = u[0] / u[2], ry = u[1] / u[2];
    if (Jq) {
        Mat<2, 3> dproj;
        dproj(0, 0) = 1.0 / u[2];
        dproj(0, 2) = -u[0] / (u[2] * u[2]);
        dproj(1, 1) = 1.0 / u[2];
        dproj(1, 2) = -u[1] / (u[2] * u[2]);
        Mat<2, 2> S;
        S(0, 0) = sic[0];
        S(1, 1) = sic[1];
        Mat<2, 3> a = (S * dproj * T.t()) * cam.q.conjugate().matrix() * hat(z_tgt_center);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) Jq[i * 3 + j] = a(i, j);
    }
    r[0] = sic[0] * rx;
    r[1] = sic[1] * ry;
}

// CeresPreIntegrationErrorFactor::Evaluate.  Ji, Jj: 15x15 w.r.t. (q local, p, v, bg, ba) of frame i / j.
// bg0/ba0: frame_i->motion.{bg,ba} as seen in USER memory (see header comment).
void eval_imu(const FrameState &fi, const FrameState &fj, const ImuFactorData &pre, const Vec3 &bg0, const Vec3 &ba0,
              const Extrinsic &imu, double r_out[15], Mat<15, 15> *Ji, Mat<15, 15> *Jj) {
    const Vec3 gravity = vec3(0.0, 0.0, -GRAVITY_NOMINAL);
    const Quat q_i = fi.q * imu.q;
    const Vec3 p_i = fi.p + fi.q * imu.p;
    const Quat q_j = fj.q * imu.q;
    const Vec3 p_j = fj.p + fj.q * imu.p;
    const double dt = pre.dt;
    const Vec3 dbg = fi.bg - bg0;
    const Vec3 dba = fi.ba - ba0;
    Mat<15, 1> r;
    Vec3 rq = logmap((pre.dq * expmap(pre.dq_dbg * dbg)).conjugate() * q_i.conjugate() * q_j);
    Vec3 rp = q_i.conjugate() * (p_j - p_i - fi.v * dt - gravity * (0.5 * dt * dt)) -
              (pre.dp + pre.dp_dbg * dbg + pre.dp_dba * dba);
    Vec3 rv = q_i.conjugate() * (fj.v - fi.v - gravity * dt) - (pre.dv + pre.dv_dbg * dbg + pre.dv_dba * dba);
    Vec3 rbg = fj.bg - fi.bg;
    Vec3 rba = fj.ba - fi.ba;
    for (int i = 0; i < 3; ++i) {
        r[ES_Q + i] = rq[i];
        r[ES_P + i] = rp[i];
        r[ES_V + i] = rv[i];
        r[ES_BG + i] = rbg[i];
        r[ES_BA + i] = rba[i];
    }
    if (Ji) {
        Mat<15, 15> J;
        Mat3 Jr_inv = inverse3(right_jacobian(rq));
        Mat3 Rqi_t = q_i.conjugate().matrix();
        Mat3 Rimu_t = imu.q.conjugate().matrix();
        // dq_i
        J.set_block<3, 3>(ES_Q, ES_Q, -(Jr_inv * q_j.conjugate().matrix() * fi.q.matrix()));
        J.set_block<3, 3>(ES_P, ES_Q,
                          Rimu_t * hat(fi.q.conjugate() * (p_j - fi.p - fi.v * dt - gravity * (0.5 * dt * dt))));
        J.set_block<3, 3>(ES_V, ES_Q, Rimu_t * hat(fi.q.conjugate() * (fj.v - fi.v - gravity * dt)));
        // dp_i
        J.set_block<3, 3>(ES_P, ES_P, -Rqi_t);
        // dv_i
        J.set_block<3, 3>(ES_P, ES_V, Rqi_t * (-dt));
        J.set_block<3, 3>(ES_V, ES_V, -Rqi_t);
        // dbg_i
        J.set_block<3, 3>(ES_Q, ES_BG,
                          -(Jr_inv * expmap(rq).conjugate().matrix() * right_jacobian(pre.dq_dbg * dbg) * pre.dq_dbg));
        J.set_block<3, 3>(ES_P, ES_BG, -pre.dp_dbg);
        J.set_block<3, 3>(ES_V, ES_BG, -pre.dv_dbg);
        J.set_block<3, 3>(ES_BG, ES_BG, -Mat3::identity());
        // dba_i
        J.set_block<3, 3>(ES_P, ES_BA, -pre.dp_dba);
        J.set_block<3, 3>(ES_V, ES_BA, -pre.dv_dba);
        J.set_block<3, 3>(ES_BA, ES_BA, -Mat3::identity());
        *Ji = pre.sqrt_inv_cov * J;
    }
    if (Jj) {
        Mat<15, 15> J;
        Mat3 Jr_inv = inverse3(right_jacobian(rq));
        Mat3 Rqi_t = q_i.conjugate().matrix();
        J.set_block<3, 3>(ES_Q, ES_Q, Jr_inv * imu.q.conjugate().matrix());
        J.set_block<3, 3>(ES_P, ES_Q, -(Rqi_t * fj.q.matrix() * hat(imu.p)));
        J.set_block<3, 3>(ES_P, ES_P, Rqi_t);
        J.set_block<3, 3>(ES_V, ES_V, Rqi_t);
        J.set_block<3, 3>(ES_BG, ES_BG, Mat3::identity());
        J.set_block<3, 3>(ES_BA, ES_BA, Mat3::identity());
        *Jj = pre.sqrt_inv_cov * J;
    }
    Mat<15, 1> rw = pre.sqrt_inv_cov * r;
    for (int i = 0; i < 15; ++i) r_out[i] = rw[i];
}

// CeresMarginalizationFactor::Evaluate: residual (n) and, optionally, the block-diagonal B
// (per prior frame, 3x3 Jr^-1(rq)); the Jacobian w.r.t. the local state is sqrt_info * B.
void eval_prior_delta(const std::vector<FrameState> &states, int prior_n, const int *prior_frames,
                      const double *prior_lin, std::vector<double> &delta, std::vector<Mat3> *Jq) {
    delta.assign((size_t)15 * prior_n, 0.0);
    if (Jq) Jq->resize(prior_n);
    for (int i = 0; i < prior_n; ++i) {
        const FrameState &s = states[prior_frames[i]];
        FrameState l = load_state(prior_lin + 16 * i);
        Vec3 rq = logmap(l.q.conjugate() * s.q);
        Vec3 rp = s.p - l.p, rv = s.v - l.v, rbg = s.bg - l.bg, rba = s.ba - l.ba;
        for (int k = 0; k < 3; ++k) {
            delta[15 * i + ES_Q + k] = rq[k];
            delta[15 * i + ES_P + k] = rp[k];
            delta[15 * i + ES_V + k] = rv[k];
            delta[15 * i + ES_BG + k] = rbg[k];
            delta[15 * i + ES_BA + k] = rba[k];
        }
        if (Jq) (*Jq)[i] = inverse3(right_jacobian(rq));
    }
}

// ------------------------------------------------------------------ the solver
struct Problem {
    const xrhip_ba_problem *P;
    int nf, nl;
    std::vector<int> pose_off, motion_off, lm_off;   // offset into the local (tangent) vector or -1
    int n_pose_motion = 0;                           // frame dofs come first, landmarks after
    int n_local = 0;
    Extrinsic cam, imu;
    std::vector<ImuFactorData> imus;
    std::vector<double> bias_ref;   // [n_imu][6]: user-state bg, ba of frame i
};

struct State {
    std::vector<FrameState> f;
    std::vector<double> d;   // inverse depths
};

void init_problem(Problem &pb, const xrhip_ba_problem *P) {
    pb.P = P;
    pb.nf = P->n_frames;
    pb.nl = P->n_landmarks;
    pb.pose_off.assign(pb.nf, -1);
    pb.motion_off.assign(pb.nf, -1);
    pb.lm_off.assign(pb.nl, -1);
    int off = 0;
    for (int f = 0; f < pb.nf; ++f) {
        if (!(P->frame_fix[f] & XRHIP_FIX_POSE)) {
            pb.pose_off[f] = off;
            off += 6;
        }
        if (!(P->frame_fix[f] & XRHIP_FIX_MOTION)) {
            pb.motion_off[f] = off;
            off += 9;
        }
    }
    pb.n_pose_motion = off;
    // only landmarks that appear in a residual block survive Ceres' preprocessing
    std::vector<char> used(pb.nl, 0);
    for (int o = 0; o < P->n_obs; ++o) used[P->obs_lm[o]] = 1;
    for (int l = 0; l < pb.nl; ++l)
        if (used[l] && !P->landmark_fix[l]) {
            pb.lm_off[l] = off;
            off += 1;
        }
    pb.n_local = off;
    pb.cam.q = Quat::from_xyzw(P->cam_q_bc);
    pb.cam.p = vec3(P->cam_p_bc[0], P->cam_p_bc[1], P->cam_p_bc[2]);
    pb.imu.q = Quat::from_xyzw(P->imu_q_bi);
    pb.imu.p = vec3(P->imu_p_bi[0], P->imu_p_bi[1], P->imu_p_bi[2]);
    pb.imus.resize(P->n_imu);
    pb.bias_ref.resize((size_t)6 * P->n_imu);
    for (int k = 0; k < P->n_imu; ++k) {
        pb.imus[k] = load_imu(P->imu_data + (size_t)XRHIP_IMU_DIM * k);
        const double *s = P->frame_state + 16 * P->imu_i[k];
        for (int i = 0; i < 6; ++i) pb.bias_ref[6 * k + i] = s[10 + i];
    }
}

// Evaluates cost (0.5 sum rho) and optionally the dense normal equations of the
// robustified, UNSCALED Jacobian: H = J^T J, g = J^T r over the local dofs.
double evaluate(const Problem &pb, const State &x, DMat *H, std::vector<double> *g) {
    const xrhip_ba_problem *P = pb.P;
    const int n = pb.n_local;
    if (H) {
        *H = DMat(n, n);
        g->assign(n, 0.0);
    }
    double cost = 0.0;
    auto add_blocks = [&](int nres, const double *r, int nb, const int *offs, const int *sizes,
                          const double *const *Js) {
        // Js[b]: nres x sizes[b] row-major
        for (int a = 0; a < nb; ++a) {
            if (offs[a] < 0) continue;
            for (int i = 0; i < sizes[a]; ++i) {
                double s = 0;
                for (int k = 0; k < nres; ++k) s += Js[a][k * sizes[a] + i] * r[k];
                (*g)[offs[a] + i] += s;
            }
            for (int b = 0; b < nb; ++b) {
                if (offs[b] < 0) continue;
                for (int i = 0; i < sizes[a]; ++i)
                    for (int j = 0; j < sizes[b]; ++j) {
                        double s = 0;
                        for (int k = 0; k < nres; ++k) s += Js[a][k * sizes[a] + i] * Js[b][k * sizes[b] + j];
                        (*H)(offs[a] + i, offs[b] + j) += s;
                    }
            }
        }
    };
    // reprojection factors (CauchyLoss(1.0): rho = log(1+s), rho' = 1/(1+s), rho'' < 0 => plain sqrt(rho') scaling)
    for (int o = 0; o < P->n_obs; ++o) {
        const int ft = P->obs_tgt[o], fr = P->obs_ref[o], l = P->obs_lm[o];
        const int ot = pb.pose_off[ft], orf = pb.pose_off[fr], ol = pb.lm_off[l];
        if (ot < 0 && orf < 0 && ol < 0) continue;   // constant residual block: removed by the preprocessor
        double r[2], Jt[12], Jr[12], Jl[2];
        Vec3 zt = vec3(P->obs_z_tgt[3 * o], P->obs_z_tgt[3 * o + 1], P->obs_z_tgt[3 * o + 2]);
        Vec3 zr = vec3(P->obs_z_ref[3 * o], P->obs_z_ref[3 * o + 1], P->obs_z_ref[3 * o + 2]);
        eval_reprojection(x.f[ft], x.f[fr], x.d[l], zt, zr, pb.cam, P->sqrt_inv_cov, r, H ? Jt : nullptr,
                          H ? Jr : nullptr, H ? Jl : nullptr);
        double s = r[0] * r[0] + r[1] * r[1];
        cost += 0.5 * std::log(1.0 + s);
        if (H) {
            double rho1 = std::max(std::numeric_limits<double>::min(), 1.0 / (1.0 + s));
            double sc = std::sqrt(rho1);
            for (int i = 0; i < 12; ++i) {
                Jt[i] *= sc;
                Jr[i] *= sc;
            }
            Jl[0] *= sc;
            Jl[1] *= sc;
            r[0] *= sc;
            r[1] *= sc;
            int offs[3] = {ot, orf, ol}, sizes[3] = {6, 6, 1};
            const double *Js[3] = {Jt, Jr, Jl};
            add_blocks(2, r, 3, offs, sizes, Js);
        }
    }
    for (int o = 0; o < P->n_rot; ++o) {
        const int ft = P->rot_tgt[o], fr = P->rot_ref[o];
        const int ot = pb.pose_off[ft];
        if (ot < 0) continue;
        double r[2], Jq[6], J6[12];
        Vec3 zt = vec3(P->rot_z_tgt[3 * o], P->rot_z_tgt[3 * o + 1], P->rot_z_tgt[3 * o + 2]);
        Vec3 zr = vec3(P->rot_z_ref[3 * o], P->rot_z_ref[3 * o + 1], P->rot_z_ref[3 * o + 2]);
        eval_rotation(x.f[ft], x.f[fr], zt, zr, pb.cam, P->sqrt_inv_cov, r, H ? Jq : nullptr);
        double s = r[0] * r[0] + r[1] * r[1];
        cost += 0.5 * std::log(1.0 + s);
        if (H) {
            double sc = std::sqrt(std::max(std::numeric_limits<double>::min(), 1.0 / (1.0 + s)));
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 6; ++j) J6[i * 6 + j] = j < 3 ? Jq[i * 3 + j] * sc : 0.0;
            r[0] *= sc;
            r[1] *= sc;
            int offs[1] = {ot}, sizes[1] = {6};
            const double *Js[1] = {J6};
            add_blocks(2, r, 1, offs, sizes, Js);
        }
    }
    // IMU factors (no loss)
    for (int k = 0; k < P->n_imu; ++k) {
        const int fi = P->imu_i[k], fj = P->imu_j[k];
        const bool act = pb.pose_off[fi] >= 0 || pb.motion_off[fi] >= 0 || pb.pose_off[fj] >= 0 ||
                         pb.motion_off[fj] >= 0;
        if (!act) continue;
        double r[15];
        Mat<15, 15> Ji, Jj;
        Vec3 bg0 = vec3(pb.bias_ref[6 * k], pb.bias_ref[6 * k + 1], pb.bias_ref[6 * k + 2]);
        Vec3 ba0 = vec3(pb.bias_ref[6 * k + 3], pb.bias_ref[6 * k + 4], pb.bias_ref[6 * k + 5]);
        eval_imu(x.f[fi], x.f[fj], pb.imus[k], bg0, ba0, pb.imu, r, H ? &Ji : nullptr, H ? &Jj : nullptr);
        double s = 0;
        for (int i = 0; i < 15; ++i) s += r[i] * r[i];
        cost += 0.5 * s;
        if (H) {
            double Jip[15 * 6], Jim[15 * 9], Jjp[15 * 6], Jjm[15 * 9];
            for (int i = 0; i < 15; ++i) {
                for (int j = 0; j < 6; ++j) {
                    Jip[i * 6 + j] = Ji(i, j);
                    Jjp[i * 6 + j] = Jj(i, j);
                }
                for (int j = 0; j < 9; ++j) {
                    Jim[i * 9 + j] = Ji(i, 6 + j);
                    Jjm[i * 9 + j] = Jj(i, 6 + j);
                }
            }
            int offs[4] = {pb.pose_off[fi], pb.motion_off[fi], pb.pose_off[fj], pb.motion_off[fj]};
            int sizes[4] = {6, 9, 6, 9};
            const double *Js[4] = {Jip, Jim, Jjp, Jjm};
            add_blocks(15, r, 4, offs, sizes, Js);
        }
    }
    // marginalisation prior (no loss)
    if (P->prior_n > 0) {
        const int np = 15 * P->prior_n;
        std::vector<double> delta;
        std::vector<Mat3> Jq;
        eval_prior_delta(x.f, P->prior_n, P->prior_frames, P->prior_lin, delta, H ? &Jq : nullptr);
        std::vector<double> r(np);
        for (int i = 0; i < np; ++i) {
            double s = P->prior_infovec[i];
            const double *row = P->prior_sqrt_info + (size_t)i * np;
            for (int j = 0; j < np; ++j) s += row[j] * delta[j];
            r[i] = s;
        }
        double s = 0;
        for (int i = 0; i < np; ++i) s += r[i] * r[i];
        cost += 0.5 * s;
        if (H) {
            // J = sqrt_info * B, B = blockdiag(Jr^-1 on the q rows, identity elsewhere)
            DMat J(np, np);
            for (int i = 0; i < np; ++i) {
                const double *row = P->prior_sqrt_info + (size_t)i * np;
                for (int f = 0; f < P->prior_n; ++f) {
                    for (int c = 0; c < 3; ++c) {
                        double v = 0;
                        for (int k = 0; k < 3; ++k) v += row[15 * f + k] * Jq[f](k, c);
                        J(i, 15 * f + c) = v;
                    }
                    for (int c = 3; c < 15; ++c) J(i, 15 * f + c) = row[15 * f + c];
                }
            }
            // column -> local offset
            std::vector<int> col(np, -1);
            for (int f = 0; f < P->prior_n; ++f) {
                int fr = P->prior_frames[f];
                for (int c = 0; c < 6; ++c)
                    if (pb.pose_off[fr] >= 0) col[15 * f + c] = pb.pose_off[fr] + c;
                for (int c = 0; c < 9; ++c)
                    if (pb.motion_off[fr] >= 0) col[15 * f + 6 + c] = pb.motion_off[fr] + c;
            }
            std::vector<double> Jtr(np, 0.0);
            for (int i = 0; i < np; ++i)
                for (int j = 0; j < np; ++j) Jtr[j] += J(i, j) * r[i];
            for (int a = 0; a < np; ++a) {
                if (col[a] < 0) continue;
                (*g)[col[a]] += Jtr[a];
            }
            for (int a = 0; a < np; ++a) {
                if (col[a] < 0) continue;
                for (int b = 0; b < np; ++b) {
                    if (col[b] < 0) continue;
                    double v = 0;
                    for (int i = 0; i < np; ++i) v += J(i, a) * J(i, b);
                    (*H)(col[a], col[b]) += v;
                }
            }
        }
    }
    return cost;
}

// Plus over all blocks: q <- (q * expmap(dq)).normalized(), everything else additive
void plus(const Problem &pb, const State &x, const std::vector<double> &delta, State &out) {
    out = x;
    for (int f = 0; f < pb.nf; ++f) {
        if (pb.pose_off[f] >= 0) {
            const double *d = &delta[pb.pose_off[f]];
            out.f[f].q = (x.f[f].q * expmap(vec3(d[0], d[1], d[2]))).normalized();
            out.f[f].p = x.f[f].p + vec3(d[3], d[4], d[5]);
        }
        if (pb.motion_off[f] >= 0) {
            const double *d = &delta[pb.motion_off[f]];
            out.f[f].v = x.f[f].v + vec3(d[0], d[1], d[2]);
            out.f[f].bg = x.f[f].bg + vec3(d[3], d[4], d[5]);
            out.f[f].ba = x.f[f].ba + vec3(d[6], d[7], d[8]);
        }
    }
    for (int l = 0; l < pb.nl; ++l)
        if (pb.lm_off[l] >= 0) out.d[l] = x.d[l] + delta[pb.lm_off[l]];
}

// ambient-space helpers over the active blocks
double ambient_norm(const Problem &pb, const State &x) {
    double s = 0;
    for (int f = 0; f < pb.nf; ++f) {
        if (pb.pose_off[f] >= 0) {
            const Quat &q = x.f[f].q;
            s += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w + x.f[f].p.squaredNorm();
        }
        if (pb.motion_off[f] >= 0) s += x.f[f].v.squaredNorm() + x.f[f].bg.squaredNorm() + x.f[f].ba.squaredNorm();
    }
    for (int l = 0; l < pb.nl; ++l)
        if (pb.lm_off[l] >= 0) s += x.d[l] * x.d[l];
    return std::sqrt(s);
}
void ambient_diff(const Problem &pb, const State &a, const State &b, double &l2, double &linf) {
    double s = 0, m = 0;
    auto acc = [&](double d) {
        s += d * d;
        m = std::max(m, std::fabs(d));
    };
    for (int f = 0; f < pb.nf; ++f) {
        if (pb.pose_off[f] >= 0) {
            acc(a.f[f].q.x - b.f[f].q.x);
            acc(a.f[f].q.y - b.f[f].q.y);
            acc(a.f[f].q.z - b.f[f].q.z);
            acc(a.f[f].q.w - b.f[f].q.w);
            for (int i = 0; i < 3; ++i) acc(a.f[f].p[i] - b.f[f].p[i]);
        }
        if (pb.motion_off[f] >= 0)
            for (int i = 0; i < 3; ++i) {
                acc(a.f[f].v[i] - b.f[f].v[i]);
                acc(a.f[f].bg[i] - b.f[f].bg[i]);
                acc(a.f[f].ba[i] - b.f[f].ba[i]);
            }
    }
    for (int l = 0; l < pb.nl; ++l)
        if (pb.lm_off[l] >= 0) acc(a.d[l] - b.d[l]);
    l2 = std::sqrt(s);
    linf = m;
}

// Solve (Hs + diag(D2)) y = gs with the landmark (scalar) blocks Schur-eliminated first.
bool schur_solve(const Problem &pb, const DMat &Hs, const std::vector<double> &gs, const std::vector<double> &D2,
                 std::vector<double> &y) {
    const int n = pb.n_local, np = pb.n_pose_motion, nl = n - np;
    DMat S(np, np);
    std::vector<double> b(np);
    for (int i = 0; i < np; ++i) {
        b[i] = gs[i];
        for (int j = 0; j < np; ++j) S(i, j) = Hs(i, j);
        S(i, i) += D2[i];
    }
    std::vector<double> hll(nl);
    for (int l = 0; l < nl; ++l) {
        const int c = np + l;
        const double d = Hs(c, c) + D2[c];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        hll[l] = d;
        const double inv = 1.0 / d;
        // only the pose columns of frames observing the landmark are non-zero
        std::vector<int> nz;
        for (int i = 0; i < np; ++i)
            if (Hs(i, c) != 0.0) nz.push_back(i);
        for (int i : nz) {
            const double wi = Hs(i, c) * inv;
            b[i] -= wi * gs[c];
            for (int j : nz) S(i, j) -= wi * Hs(j, c);
        }
    }
    DMat L;
    if (np > 0) {
        if (!cholesky_lower(S, L)) return false;
        // forward / backward substitution
        for (int i = 0; i < np; ++i) {
            double s = b[i];
            for (int k = 0; k < i; ++k) s -= L(i, k) * b[k];
            b[i] = s / L(i, i);
        }
        for (int i = np - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < np; ++k) s -= L(k, i) * b[k];
            b[i] = s / L(i, i);
        }
    }
    y.assign(n, 0.0);
    for (int i = 0; i < np; ++i) y[i] = b[i];
    for (int l = 0; l < nl; ++l) {
        const int c = np + l;
        double s = gs[c];
        for (int i = 0; i < np; ++i) s -= Hs(i, c) * y[i];
        y[c] = s / hll[l];
    }
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(y[i])) return false;
    return true;
}

struct Dogleg {
    double radius = 1e4;
    double mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0, mu_increase = 10.0;
    bool reuse = false;
    std::vector<double> diagonal, gradient, gn;   // in the Jacobi-scaled space
    double alpha = 0, step_norm = 0;
};

}   // namespace

extern "C" {

// Optional per-iteration record of the last orc_ba_solve_trace call (regression pin of the minimiser's decisions,
// tests/golden/ba_snapshots): rows of {iteration, x_cost, candidate cost, model cost change, relative decrease,
// radius before the decision, |step|, mu, accepted}.
static thread_local double *g_trace = nullptr;   // per thread: several pipelines may solve at once (instances, backend threads)
static thread_local int g_trace_cap = 0, g_trace_rows = 0;
static int orc_ba_solve_impl(const xrhip_ba_problem *P, xrhip_ba_summary *summary);
int orc_ba_solve(const xrhip_ba_problem *P, xrhip_ba_summary *summary) {
    g_trace = nullptr;
    g_trace_cap = g_trace_rows = 0;
    return orc_ba_solve_impl(P, summary);
}
int orc_ba_solve_trace(const xrhip_ba_problem *P, xrhip_ba_summary *summary, double *trace9, int max_rows, int *rows) {
    g_trace = trace9;
    g_trace_cap = max_rows;
    g_trace_rows = 0;
    const int rc = orc_ba_solve_impl(P, summary);
    if (rows) *rows = g_trace_rows;
    g_trace = nullptr;
    return rc;
}
static int orc_ba_solve_impl(const xrhip_ba_problem *P, xrhip_ba_summary *summary) {
    Problem pb;
    init_problem(pb, P);
    const int n = pb.n_local;
    State x;
    x.f.resize(pb.nf);
    for (int f = 0; f < pb.nf; ++f) x.f[f] = load_state(P->frame_state + 16 * f);
    x.d.assign(P->inv_depth, P->inv_depth + pb.nl);
    xrhip_ba_summary sm;
    std::memset(&sm, 0, sizeof(sm));
    if (n == 0) {
        sm.termination = XRHIP_BA_CONVERGENCE;
        sm.usable = 1;
        if (summary) *summary = sm;
        return 0;
    }
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32, max_radius = 1e16;
    const int max_invalid = 5;

    DMat H;
    std::vector<double> g;
    double x_cost = evaluate(pb, x, &H, &g);
    sm.initial_cost = x_cost;
    // Jacobi scaling from the initial Jacobian: 1 / (1 + ||col||)
    std::vector<double> scale(n);
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H(i, i)));
    double x_norm = ambient_norm(pb, x);
    auto gradient_max_norm = [&](const State &xs, const std::vector<double> &grad) {
        std::vector<double> neg(n);
        for (int i = 0; i < n; ++i) neg[i] = -grad[i];
        State pg;
        plus(pb, xs, neg, pg);
        double l2, linf;
        ambient_diff(pb, xs, pg, l2, linf);
        return linf;
    };
    double gmax = gradient_max_norm(x, g);
    Dogleg dl;
    int iteration = 0, invalid = 0;
    bool step_ok = true;   // iteration 0 counts as successful
    State best = x;
    double minimum_cost = x_cost;
    int termination = XRHIP_BA_NO_CONVERGENCE;
    bool failure = false;
    DMat Hs(n, n);
    std::vector<double> gs(n);
    auto rescale = [&]() {
        for (int i = 0; i < n; ++i) {
            gs[i] = g[i] * scale[i];
            for (int j = 0; j < n; ++j) Hs(i, j) = H(i, j) * scale[i] * scale[j];
        }
    };
    rescale();
    auto refresh_user_state = [&]() {   // StateUpdatingCallback after a successful iteration
        for (int k = 0; k < P->n_imu; ++k) {
            const FrameState &s = best.f[P->imu_i[k]];
            for (int i = 0; i < 3; ++i) {
                pb.bias_ref[6 * k + i] = s.bg[i];
                pb.bias_ref[6 * k + 3 + i] = s.ba[i];
            }
        }
    };
    while (true) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (step_ok) {
            sm.successful_steps += (iteration > 0);
            if (x_cost < minimum_cost || iteration == 0) {
                minimum_cost = x_cost;
                best = x;
            }
            refresh_user_state();
        }
        if (iteration >= P->max_iterations) {
            termination = XRHIP_BA_NO_CONVERGENCE;
            break;
        }
        if (step_ok && gmax <= gradient_tolerance) {
            termination = XRHIP_BA_CONVERGENCE;
            break;
        }
        if (dl.radius <= min_trust_region_radius) {
            termination = XRHIP_BA_CONVERGENCE;
            break;
        }
        ++iteration;
        step_ok = false;
        // ---- ComputeTrustRegionStep (DoglegStrategy::ComputeStep)
        bool linear_ok = true;
        std::vector<double> step(n, 0.0);
        auto traditional_dogleg = [&]() {
            const double gnorm = std::sqrt([&] { double s = 0; for (double v : dl.gradient) s += v * v; return s; }());
            const double gn_norm = std::sqrt([&] { double s = 0; for (double v : dl.gn) s += v * v; return s; }());
            if (gn_norm <= dl.radius) {
                for (int i = 0; i < n; ++i) step[i] = dl.gn[i] / dl.diagonal[i];
                dl.step_norm = gn_norm;
                return;
            }
            if (gnorm * dl.alpha >= dl.radius) {
                for (int i = 0; i < n; ++i) step[i] = (-(dl.radius / gnorm) * dl.gradient[i]) / dl.diagonal[i];
                dl.step_norm = dl.radius;
                return;
            }
            double gdot = 0;
            for (int i = 0; i < n; ++i) gdot += dl.gradient[i] * dl.gn[i];
            const double b_dot_a = -dl.alpha * gdot;
            const double a_sq = std::pow(dl.alpha * gnorm, 2.0);
            const double bma_sq = a_sq - 2 * b_dot_a + std::pow(gn_norm, 2);
            const double c = b_dot_a - a_sq;
            const double d = std::sqrt(c * c + bma_sq * (std::pow(dl.radius, 2.0) - a_sq));
            const double beta = (c <= 0) ? (d - c) / bma_sq : (dl.radius * dl.radius - a_sq) / (d + c);
            double sn = 0;
            for (int i = 0; i < n; ++i) {
                double v = (-dl.alpha * (1.0 - beta)) * dl.gradient[i] + beta * dl.gn[i];
                sn += v * v;
                step[i] = v / dl.diagonal[i];
            }
            dl.step_norm = std::sqrt(sn);
        };
        if (dl.reuse) {
            traditional_dogleg();
        } else {
            dl.reuse = true;
            dl.diagonal.resize(n);
            dl.gradient.resize(n);
            for (int i = 0; i < n; ++i)
                dl.diagonal[i] = std::sqrt(std::min(std::max(Hs(i, i), 1e-6), 1e32));
            for (int i = 0; i < n; ++i) dl.gradient[i] = gs[i] / dl.diagonal[i];
            // Cauchy point: alpha = |gradient|^2 / |J (D^-1 gradient)|^2
            {
                std::vector<double> sg(n);
                for (int i = 0; i < n; ++i) sg[i] = dl.gradient[i] / dl.diagonal[i];
                double jg2 = 0;
                for (int i = 0; i < n; ++i) {
                    double s = 0;
                    for (int j = 0; j < n; ++j) s += Hs(i, j) * sg[j];
                    jg2 += sg[i] * s;
                }
                double g2 = 0;
                for (double v : dl.gradient) g2 += v * v;
                dl.alpha = g2 / jg2;
            }
            // Gauss-Newton step with mu * diagonal^2 regularisation
            linear_ok = false;
            while (dl.mu < dl.max_mu) {
                std::vector<double> D2(n), y;
                for (int i = 0; i < n; ++i) D2[i] = dl.diagonal[i] * dl.diagonal[i] * dl.mu;
                if (schur_solve(pb, Hs, gs, D2, y)) {
                    dl.gn.resize(n);
                    for (int i = 0; i < n; ++i) dl.gn[i] = -dl.diagonal[i] * y[i];
                    linear_ok = true;
                    break;
                }
                dl.mu *= dl.mu_increase;
            }
            if (linear_ok) traditional_dogleg();
        }
        double model_cost_change = 0;
        bool valid = false;
        if (linear_ok) {
            // -(J step)^T (r + J step / 2) = -step^T gs - 0.5 step^T Hs step   (scaled space)
            double sg = 0, shs = 0;
            for (int i = 0; i < n; ++i) {
                sg += step[i] * gs[i];
                double s = 0;
                for (int j = 0; j < n; ++j) s += Hs(i, j) * step[j];
                shs += step[i] * s;
            }
            model_cost_change = -sg - 0.5 * shs;
            valid = model_cost_change > 0.0;
        }
        if (!valid) {
            if (++invalid >= max_invalid) {
                failure = true;
                termination = XRHIP_BA_FAILURE;
                break;
            }
            dl.mu *= dl.mu_increase;   // StepIsInvalid
            dl.reuse = false;
            continue;
        }
        invalid = 0;
        std::vector<double> delta(n);
        for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
        State cand;
        plus(pb, x, delta, cand);
        double cand_cost = evaluate(pb, cand, nullptr, nullptr);
        if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
        // ParameterToleranceReached
        double step_l2, step_linf;
        ambient_diff(pb, x, cand, step_l2, step_linf);
        if (step_l2 <= parameter_tolerance * (x_norm + parameter_tolerance)) {
            termination = XRHIP_BA_CONVERGENCE;
            break;
        }
        // FunctionToleranceReached
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= function_tolerance * x_cost) {
            termination = XRHIP_BA_CONVERGENCE;
            break;
        }
        const double relative_decrease = cost_change / model_cost_change;
        if (std::getenv("ORC_BA_TRACE"))
            std::fprintf(stderr, "it %2d cost %.9e cand %.9e model %.3e rho %.3f radius %.3e |step| %.3e mu %.1e\n",
                         iteration, x_cost, cand_cost, model_cost_change, relative_decrease, dl.radius, step_l2, dl.mu);
        if (g_trace && g_trace_rows < g_trace_cap) {
            double *row = g_trace + 9 * (g_trace_rows++);
            row[0] = iteration; row[1] = x_cost; row[2] = cand_cost; row[3] = model_cost_change; row[4] = relative_decrease;
            row[5] = dl.radius; row[6] = step_l2; row[7] = dl.mu; row[8] = relative_decrease > min_relative_decrease ? 1.0 : 0.0;
        }
        if (relative_decrease > min_relative_decrease) {
            // HandleSuccessfulStep
            x = cand;
            x_norm = ambient_norm(pb, x);
            x_cost = evaluate(pb, x, &H, &g);   // user state (bias_ref) still the old one here
            rescale();
            gmax = gradient_max_norm(x, g);
            step_ok = true;
            if (relative_decrease < 0.25) dl.radius *= 0.5;
            if (relative_decrease > 0.75) dl.radius = std::max(dl.radius, 3.0 * dl.step_norm);
            dl.radius = std::min(dl.radius, max_radius);
            dl.mu = std::max(dl.min_mu, 2.0 * dl.mu / dl.mu_increase);
            dl.reuse = false;
        } else {
            dl.radius *= 0.5;   // StepRejected
            dl.reuse = true;
        }
    }
    (void)failure;
    sm.iterations = iteration;
    sm.termination = termination;
    sm.usable = termination != XRHIP_BA_FAILURE;
    sm.final_cost = minimum_cost;
    for (int f = 0; f < pb.nf; ++f) store_state(best.f[f], P->frame_state + 16 * f);
    for (int l = 0; l < pb.nl; ++l) P->inv_depth[l] = best.d[l];
    if (summary) *summary = sm;
    return 0;
}

// PreIntegrator::integrate
int orc_preintegrate(const double *samples, int n, double t_end, const double *bg, const double *ba,
                     const double *noise36, int jac, int cv, double *out) {
    if (n <= 0) return -1;
    PreInt pi;
    for (int i = 0; i < 9; ++i) {
        pi.cov_w.a[i] = noise36[i];
        pi.cov_a.a[i] = noise36[9 + i];
        pi.cov_bg.a[i] = noise36[18 + i];
        pi.cov_ba.a[i] = noise36[27 + i];
    }
    Vec3 vbg = vec3(bg[0], bg[1], bg[2]), vba = vec3(ba[0], ba[1], ba[2]);
    for (int i = 0; i + 1 < n; ++i) {
        const double *d = samples + 7 * i;
        pi.increment(samples[7 * (i + 1)] - d[0], vec3(d[1], d[2], d[3]), vec3(d[4], d[5], d[6]), vbg, vba, jac, cv);
    }
    const double *d = samples + 7 * (n - 1);
    pi.increment(t_end - d[0], vec3(d[1], d[2], d[3]), vec3(d[4], d[5], d[6]), vbg, vba, jac, cv);
    if (cv) {   // compute_sqrt_inv_cov: LLT(cov^-1).matrixL().transpose()
        DMat c(15, 15);
        for (int i = 0; i < 15; ++i)
            for (int j = 0; j < 15; ++j) c(i, j) = pi.cov(i, j);
        if (!invert(c)) return -2;
        DMat L;
        if (!cholesky_lower(c, L)) return -3;
        for (int i = 0; i < 15; ++i)
            for (int j = 0; j < 15; ++j) pi.sqrt_inv_cov(i, j) = L(j, i);
    }
    out[0] = pi.t;
    pi.q.to_xyzw(out + 1);
    for (int i = 0; i < 3; ++i) {
        out[5 + i] = pi.p[i];
        out[8 + i] = pi.v[i];
    }
    const Mat3 *ms[5] = {&pi.dq_dbg, &pi.dp_dbg, &pi.dp_dba, &pi.dv_dbg, &pi.dv_dba};
    for (int k = 0; k < 5; ++k)
        for (int i = 0; i < 9; ++i) out[11 + 9 * k + i] = ms[k]->a[i];
    for (int i = 0; i < 225; ++i) out[56 + i] = pi.sqrt_inv_cov.a[i];
    return 0;
}
/* raw covariance for tests */
int orc_preintegrate_cov(const double *samples, int n, double t_end, const double *bg, const double *ba,
                         const double *noise36, double *cov225) {
    PreInt pi;
    for (int i = 0; i < 9; ++i) {
        pi.cov_w.a[i] = noise36[i];
        pi.cov_a.a[i] = noise36[9 + i];
        pi.cov_bg.a[i] = noise36[18 + i];
        pi.cov_ba.a[i] = noise36[27 + i];
    }
    Vec3 vbg = vec3(bg[0], bg[1], bg[2]), vba = vec3(ba[0], ba[1], ba[2]);
    for (int i = 0; i + 1 < n; ++i) {
        const double *d = samples + 7 * i;
        pi.increment(samples[7 * (i + 1)] - d[0], vec3(d[1], d[2], d[3]), vec3(d[4], d[5], d[6]), vbg, vba, true, true);
    }
    const double *d = samples + 7 * (n - 1);
    pi.increment(t_end - d[0], vec3(d[1], d[2], d[3]), vec3(d[4], d[5], d[6]), vbg, vba, true, true);
    for (int i = 0; i < 225; ++i) cov225[i] = pi.cov.a[i];
    return 0;
}

/* PreIntegrator::predict (preintegrator.cpp:102-112): state_j from state_i and the delta in imu_data */
void orc_predict(const double *state_i, const double *imu_data, double *state_j) {
    FrameState a = load_state(state_i), b = a;
    ImuFactorData d = load_imu(imu_data);
    const Vec3 gravity = vec3(0, 0, -GRAVITY_NOMINAL);
    b.v = a.v + gravity * d.dt + a.q * d.dv;
    b.p = a.p + gravity * (0.5 * d.dt * d.dt) + a.v * d.dt + a.q * d.dp;
    b.q = a.q * d.dq;
    store_state(b, state_j);
}

/* factor-level entry points for the finite-difference tests */
void orc_eval_reprojection(const double *st_tgt, const double *st_ref, double inv_depth, const double *z_tgt,
                           const double *z_ref, const double *cam7, const double *sic2, double *r2, double *Jt12,
                           double *Jr12, double *Jl2) {
    Extrinsic cam{Quat::from_xyzw(cam7), vec3(cam7[4], cam7[5], cam7[6])};
    eval_reprojection(load_state(st_tgt), load_state(st_ref), inv_depth, vec3(z_tgt[0], z_tgt[1], z_tgt[2]),
                      vec3(z_ref[0], z_ref[1], z_ref[2]), cam, sic2, r2, Jt12, Jr12, Jl2);
}
void orc_eval_rotation(const double *st_tgt, const double *st_ref, const double *z_tgt, const double *z_ref,
                       const double *cam7, const double *sic2, double *r2, double *Jq6) {
    Extrinsic cam{Quat::from_xyzw(cam7), vec3(cam7[4], cam7[5], cam7[6])};
    eval_rotation(load_state(st_tgt), load_state(st_ref), vec3(z_tgt[0], z_tgt[1], z_tgt[2]),
                  vec3(z_ref[0], z_ref[1], z_ref[2]), cam, sic2, r2, Jq6);
}
void orc_eval_imu(const double *st_i, const double *st_j, const double *imu_data, const double *bias_ref6,
                  const double *imu7, double *r15, double *Ji225, double *Jj225) {
    Extrinsic imu{Quat::from_xyzw(imu7), vec3(imu7[4], imu7[5], imu7[6])};
    Mat<15, 15> Ji, Jj;
    eval_imu(load_state(st_i), load_state(st_j), load_imu(imu_data), vec3(bias_ref6[0], bias_ref6[1], bias_ref6[2]),
             vec3(bias_ref6[3], bias_ref6[4], bias_ref6[5]), imu, r15, Ji225 ? &Ji : nullptr, Jj225 ? &Jj : nullptr);
    if (Ji225) std::memcpy(Ji225, Ji.a, sizeof(double) * 225);
    if (Jj225) std::memcpy(Jj225, Jj.a, sizeof(double) * 225);
}
/* cost + dense normal equations at the current state (for cross-checks); H is n_local^2, returns n_local.
 * offsets: pose_off[nf], motion_off[nf], lm_off[nl] */
int orc_ba_linearize(const xrhip_ba_problem *P, double *cost, double *H, double *g, int *pose_off, int *motion_off,
                     int *lm_off) {
    Problem pb;
    init_problem(pb, P);
    State x;
    x.f.resize(pb.nf);
    for (int f = 0; f < pb.nf; ++f) x.f[f] = load_state(P->frame_state + 16 * f);
    x.d.assign(P->inv_depth, P->inv_depth + pb.nl);
    if (pose_off) std::memcpy(pose_off, pb.pose_off.data(), sizeof(int) * pb.nf);
    if (motion_off) std::memcpy(motion_off, pb.motion_off.data(), sizeof(int) * pb.nf);
    if (lm_off && pb.nl) std::memcpy(lm_off, pb.lm_off.data(), sizeof(int) * pb.nl);
    if (H) {
        DMat Hm;
        std::vector<double> gv;
        *cost = evaluate(pb, x, &Hm, &gv);
        std::memcpy(H, Hm.a.data(), sizeof(double) * Hm.a.size());
        std::memcpy(g, gv.data(), sizeof(double) * gv.size());
    } else {
        *cost = evaluate(pb, x, nullptr, nullptr);
    }
    return pb.n_local;
}
/* Plus on a single state (QuaternionParameterization + additive blocks): delta15 = dq3,dp,dv,dbg,dba */
void orc_state_plus(const double *state, const double *delta15, double *out) {
    FrameState s = load_state(state);
    s.q = (s.q * expmap(vec3(delta15[0], delta15[1], delta15[2]))).normalized();
    s.p = s.p + vec3(delta15[3], delta15[4], delta15[5]);
    s.v = s.v + vec3(delta15[6], delta15[7], delta15[8]);
    s.bg = s.bg + vec3(delta15[9], delta15[10], delta15[11]);
    s.ba = s.ba + vec3(delta15[12], delta15[13], delta15[14]);
    store_state(s, out);
}

// CeresMarginalizationFactor::marginalize
int orc_ba_marginalize(const xrhip_marg_problem *M, double *out_sqrt_info, double *out_infovec, double *out_lin) {
    const int K = M->n_frames, N = 15 * K;
    std::vector<FrameState> st(K);
    for (int f = 0; f < K; ++f) st[f] = load_state(M->frame_state + 16 * f);
    Extrinsic cam{Quat::from_xyzw(M->cam_q_bc), vec3(M->cam_p_bc[0], M->cam_p_bc[1], M->cam_p_bc[2])};
    Extrinsic imu{Quat::from_xyzw(M->imu_q_bi), vec3(M->imu_p_bi[0], M->imu_p_bi[1], M->imu_p_bi[2])};
    // victim goes last (marginalization_factor.h:95-105)
    std::vector<int> fidx(K);
    for (int i = 0; i < K; ++i) fidx[i] = i < M->victim ? i : (i > M->victim ? i - 1 : K - 1);
    DMat H(N, N);
    std::vector<double> b(N, 0.0);
    // old prior
    if (M->prior_n > 0) {
        const int np = 15 * M->prior_n;
        std::vector<double> delta;
        std::vector<Mat3> Jq;
        eval_prior_delta(st, M->prior_n, M->prior_frames, M->prior_lin, delta, &Jq);
        std::vector<double> r(np);
        for (int i = 0; i < np; ++i) {
            double s = M->prior_infovec[i];
            for (int j = 0; j < np; ++j) s += M->prior_sqrt_info[(size_t)i * np + j] * delta[j];
            r[i] = s;
        }
        DMat J(np, np);
        for (int i = 0; i < np; ++i) {
            const double *row = M->prior_sqrt_info + (size_t)i * np;
            for (int f = 0; f < M->prior_n; ++f) {
                for (int c = 0; c < 3; ++c) {
                    double v = 0;
                    for (int k = 0; k < 3; ++k) v += row[15 * f + k] * Jq[f](k, c);
                    J(i, 15 * f + c) = v;
                }
                for (int c = 3; c < 15; ++c) J(i, 15 * f + c) = row[15 * f + c];
            }
        }
        std::vector<int> col(np);
        for (int f = 0; f < M->prior_n; ++f)
            for (int c = 0; c < 15; ++c) col[15 * f + c] = 15 * fidx[M->prior_frames[f]] + c;
        for (int a = 0; a < np; ++a) {
            double s = 0;
            for (int i = 0; i < np; ++i) s += J(i, a) * r[i];
            b[col[a]] += s;
            for (int c = 0; c < np; ++c) {
                double v = 0;
                for (int i = 0; i < np; ++i) v += J(i, a) * J(i, c);
                H(col[a], col[c]) += v;
            }
        }
    }
    // IMU factors adjacent to the victim
    for (int k = 0; k < M->n_imu; ++k) {
        const int fi = M->imu_i[k], fj = M->imu_j[k];
        ImuFactorData pre = load_imu(M->imu_data + (size_t)XRHIP_IMU_DIM * k);
        double r[15];
        Mat<15, 15> Ji, Jj;
        eval_imu(st[fi], st[fj], pre, st[fi].bg, st[fi].ba, imu, r, &Ji, &Jj);
        const int oi = 15 * fidx[fi], oj = 15 * fidx[fj];
        for (int a = 0; a < 15; ++a) {
            double si = 0, sj = 0;
            for (int i = 0; i < 15; ++i) {
                si += Ji(i, a) * r[i];
                sj += Jj(i, a) * r[i];
            }
            b[oi + a] += si;
            b[oj + a] += sj;
            for (int c = 0; c < 15; ++c) {
                double vii = 0, vij = 0, vji = 0, vjj = 0;
                for (int i = 0; i < 15; ++i) {
                    vii += Ji(i, a) * Ji(i, c);
                    vij += Ji(i, a) * Jj(i, c);
                    vji += Jj(i, a) * Ji(i, c);
                    vjj += Jj(i, a) * Jj(i, c);
                }
                H(oi + a, oi + c) += vii;
                H(oi + a, oj + c) += vij;
                H(oj + a, oi + c) += vji;
                H(oj + a, oj + c) += vjj;
            }
        }
    }
    // reprojection factors (no loss) + per-landmark info
    struct LInfo {
        double mat = 0, vec = 0;
        std::map<int, Mat<1, 6>> h;
    };
    std::vector<LInfo> linfo(M->n_landmarks);
    for (int o = 0; o < M->n_obs; ++o) {
        const int ft = M->obs_tgt[o], fr = M->obs_ref[o], l = M->obs_lm[o];
        double r[2], Jt[12], Jr[12], Jl[2];
        eval_reprojection(st[ft], st[fr], M->inv_depth[l], vec3(M->obs_z_tgt[3 * o], M->obs_z_tgt[3 * o + 1], M->obs_z_tgt[3 * o + 2]),
                          vec3(M->obs_z_ref[3 * o], M->obs_z_ref[3 * o + 1], M->obs_z_ref[3 * o + 2]), cam,
                          M->sqrt_inv_cov, r, Jt, Jr, Jl);
        const int ot = 15 * fidx[ft], orf = 15 * fidx[fr];
        for (int a = 0; a < 6; ++a) {
            b[ot + a] += Jt[a] * r[0] + Jt[6 + a] * r[1];
            b[orf + a] += Jr[a] * r[0] + Jr[6 + a] * r[1];
            for (int c = 0; c < 6; ++c) {
                H(ot + a, ot + c) += Jt[a] * Jt[c] + Jt[6 + a] * Jt[6 + c];
                H(orf + a, ot + c) += Jr[a] * Jt[c] + Jr[6 + a] * Jt[6 + c];
                H(ot + a, orf + c) += Jt[a] * Jr[c] + Jt[6 + a] * Jr[6 + c];
                H(orf + a, orf + c) += Jr[a] * Jr[c] + Jr[6 + a] * Jr[6 + c];
            }
        }
        LInfo &li = linfo[l];
        li.mat += Jl[0] * Jl[0] + Jl[1] * Jl[1];
        li.vec += Jl[0] * r[0] + Jl[1] * r[1];
        Mat<1, 6> &ht = li.h[fidx[ft]];
        Mat<1, 6> &hr = li.h[fidx[fr]];
        for (int a = 0; a < 6; ++a) {
            ht[a] += Jl[0] * Jt[a] + Jl[1] * Jt[6 + a];
            hr[a] += Jl[0] * Jr[a] + Jl[1] * Jr[6 + a];
        }
    }
    for (int l = 0; l < M->n_landmarks; ++l) {
        const LInfo &li = linfo[l];
        if (li.h.empty()) continue;
        const double inv = 1.0 / li.mat;
        if (!std::isfinite(inv)) continue;
        for (const auto &hi : li.h) {
            for (const auto &hj : li.h)
                for (int a = 0; a < 6; ++a)
                    for (int c = 0; c < 6; ++c) H(15 * hi.first + a, 15 * hj.first + c) -= hi.second[a] * inv * hj.second[c];
            for (int a = 0; a < 6; ++a) b[15 * hi.first + a] -= hi.second[a] * inv * li.vec;
        }
    }
    // Schur-eliminate the victim (last 15)
    const int R = N - 15;
    DMat Hvv(15, 15);
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) Hvv(i, j) = H(R + i, R + j);
    if (!invert(Hvv)) return -1;
    DMat Hr(R, R);
    std::vector<double> br(R);
    // T = H_rv * Hvv^-1
    DMat T(R, 15);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < 15; ++j) {
            double s = 0;
            for (int k = 0; k < 15; ++k) s += H(i, R + k) * Hvv(k, j);
            T(i, j) = s;
        }
    for (int i = 0; i < R; ++i) {
        double s = b[i];
        for (int k = 0; k < 15; ++k) s -= T(i, k) * b[R + k];
        br[i] = s;
        for (int j = 0; j < R; ++j) {
            double v = H(i, j);
            for (int k = 0; k < 15; ++k) v -= T(i, k) * H(R + k, j);
            Hr(i, j) = v;
        }
    }
    // eigen-decomposition, clamp, sqrt
    std::vector<double> w;
    DMat V;
    // symmetrise (SelfAdjointEigenSolver reads the lower triangle only)
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < i; ++j) Hr(j, i) = Hr(i, j);
    sym_eigen(Hr, w, V);
    if (std::getenv("ORC_MARG_TRACE")) {
        int nsmall = 0;
        for (int i = 0; i < R; ++i) nsmall += !(w[i] > 1.0e-8);
        std::fprintf(stderr, "marg R=%d lambda_min=%.3e lambda_2=%.3e lambda_max=%.3e  clamped=%d\n", R, w[0], w[1], w[R - 1], nsmall);
    }
    for (int i = 0; i < R; ++i) {
        const double lam = w[i] > 1.0e-8 ? w[i] : 0.0;
        const double lam_inv = w[i] > 1.0e-8 ? 1.0 / w[i] : 0.0;
        const double sl = std::sqrt(lam), sli = std::sqrt(lam_inv);
        double s = 0;
        for (int j = 0; j < R; ++j) {
            out_sqrt_info[(size_t)i * R + j] = sl * V(j, i);
            s += V(j, i) * br[j];
        }
        out_infovec[i] = sli * s;
    }
    int j = 0;
    for (int i = 0; i < K; ++i) {
        if (i == M->victim) continue;
        std::memcpy(out_lin + 16 * j, M->frame_state + 16 * i, sizeof(double) * 16);
        ++j;
    }
    return 0;
}

}   // extern "C"
