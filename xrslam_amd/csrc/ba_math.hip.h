// ba_math.hip.h -- double-precision SO(3)/factor math for the gfx950 BA kernels.
//
// Device-side restatement of the reference's factor arithmetic
// (/root/reference/xrslam/src/xrslam):
//   geometry/lie_algebra.h:8-23, lie_algebra.cpp:5-56      hat / expmap / logmap / right_jacobian / s2 basis
//   estimation/ceres/reprojection_factor.h:25-90            reprojection residual + Jacobians
//   estimation/ceres/rotation_factor.h:23-59                rotation prior
//   estimation/ceres/preintegration_factor.h:20-159         IMU residual + Jacobians
//   estimation/ceres/quaternion_parameterization.h:11-17    Plus
// All values are f64 (the reference's estimation stack is double throughout).
#pragma once
#include <hip/hip_runtime.h>

namespace xrhip {

struct V3 {
    double x, y, z;
};
struct Q4 {   // x,y,z,w storage like Eigen coeffs()
    double x, y, z, w;
};
struct M3 {   // row-major
    double m[9];
};

#define XD __host__ __device__ __forceinline__

XD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
XD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
XD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
XD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
XD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
XD V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
XD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
XD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
XD double norm(V3 a) { return sqrt(dot(a, a)); }
XD double get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
XD V3 normalized(V3 v) {
    double n2 = dot(v, v);
    return n2 > 0 ? v / sqrt(n2) : v;
}
XD V3 stable_normalized(V3 v) {
    double w = fmax(fabs(v.x), fmax(fabs(v.y), fabs(v.z)));
    V3 s = v / w;
    double z = dot(s, s);
    if (z > 0) return v / (sqrt(z) * w);
    return v;
}

XD M3 m3_identity() {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return r;
}
XD M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
XD V3 operator*(const M3 &a, V3 v) {
    return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
              a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
XD M3 operator*(const M3 &a, double s) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] * s;
    return r;
}
XD M3 operator+(const M3 &a, const M3 &b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i];
    return r;
}
XD M3 operator-(const M3 &a, const M3 &b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i];
    return r;
}
XD M3 operator-(const M3 &a) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = -a.m[i];
    return r;
}
XD M3 transpose(const M3 &a) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * j + i];
    return r;
}
XD M3 hat(V3 w) {
    M3 r;
    r.m[0] = 0;
    r.m[1] = -w.z;
    r.m[2] = w.y;
    r.m[3] = w.z;
    r.m[4] = 0;
    r.m[5] = -w.x;
    r.m[6] = -w.y;
    r.m[7] = w.x;
    r.m[8] = 0;
    return r;
}
XD M3 inverse3(const M3 &a) {
    M3 c;
    c.m[0] = a.m[4] * a.m[8] - a.m[5] * a.m[7];
    c.m[1] = a.m[2] * a.m[7] - a.m[1] * a.m[8];
    c.m[2] = a.m[1] * a.m[5] - a.m[2] * a.m[4];
    c.m[3] = a.m[5] * a.m[6] - a.m[3] * a.m[8];
    c.m[4] = a.m[0] * a.m[8] - a.m[2] * a.m[6];
    c.m[5] = a.m[2] * a.m[3] - a.m[0] * a.m[5];
    c.m[6] = a.m[3] * a.m[7] - a.m[4] * a.m[6];
    c.m[7] = a.m[1] * a.m[6] - a.m[0] * a.m[7];
    c.m[8] = a.m[0] * a.m[4] - a.m[1] * a.m[3];
    double det = a.m[0] * c.m[0] + a.m[1] * c.m[3] + a.m[2] * c.m[6];
    return c * (1.0 / det);
}

XD Q4 q_conj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
XD Q4 q_mul(Q4 a, Q4 b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
XD Q4 q_normalized(Q4 q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
}
XD V3 q_rot(Q4 q, V3 v) {
    V3 u = v3(q.x, q.y, q.z);
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + uv * q.w + cross(u, uv);
}
XD M3 q_mat(Q4 q) {
    M3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r.m[0] = 1 - (tyy + tzz);
    r.m[1] = txy - twz;
    r.m[2] = txz + twy;
    r.m[3] = txy + twz;
    r.m[4] = 1 - (txx + tzz);
    r.m[5] = tyz - twx;
    r.m[6] = txz - twy;
    r.m[7] = tyz + twx;
    r.m[8] = 1 - (txx + tyy);
    return r;
}
// sine and cosine of one angle.  On the device one sincos() call: the library's sin(), cos() and sincos() share the argument
// reduction and the two polynomials, so the values are those of the separate calls at less than half their instructions
// (tools/latency.hip: 168 ns against 365 ns on one lane) -- and these kernels are bound by instruction issue.
XD void sin_cos(double a, double &s, double &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(a, &s, &c);
#else
    s = sin(a);
    c = cos(a);
#endif
}
XD Q4 expmap(V3 w) {
    double angle = norm(w);
    V3 axis = stable_normalized(w);
    double ha = 0.5 * angle;
    double s, c;
    sin_cos(ha, s, c);
    return Q4{s * axis.x, s * axis.y, s * axis.z, c};
}
XD V3 logmap(Q4 q) {
    V3 v = v3(q.x, q.y, q.z);
    double n = norm(v);
    if (n < 2.220446049250313e-16) {
        double m = fmax(fabs(v.x), fmax(fabs(v.y), fabs(v.z)));
        n = (m > 0) ? m * norm(v / m) : 0.0;
    }
    if (n != 0.0) {
        double angle = 2.0 * atan2(n, fabs(q.w));
        if (q.w < 0) n = -n;
        return (v / n) * angle;
    }
    return v3(0, 0, 0);
}
XD M3 right_jacobian(V3 w) {
    const double root2_eps = 1.4901161193847656e-08;   // sqrt(eps)
    const double root4_eps = 1.220703125e-04;          // sqrt(sqrt(eps))
    const double qdrt720 = 5.180044506382739;          // 720^(1/4)
    const double qdrt5040 = 8.425996210694382;         // 5040^(1/4)
    const double sqrt24 = 4.898979485566356;
    const double sqrt120 = 10.954451150103322;
    double angle = norm(w);
    double cangle, sangle;
    sin_cos(angle, sangle, cangle);
    double angle2 = angle * angle;
    double cos_term, sin_term;
    if (angle > root4_eps * qdrt720) {
        cos_term = (1 - cangle) / angle2;
    } else {
        cos_term = 0.5;
        if (angle > root2_eps * sqrt24) cos_term -= angle2 / 24.0;
    }
    if (angle > root4_eps * qdrt5040) {
        sin_term = (angle - sangle) / (angle * angle2);
    } else {
        sin_term = 1.0 / 6.0;
        if (angle > root2_eps * sqrt120) sin_term -= angle2 / 120.0;
    }
    M3 hw = hat(w);
    return m3_identity() - hw * cos_term + (hw * hw) * sin_term;
}
XD void s2_tangential_basis(V3 x, V3 &b1, V3 &b2) {
    int d = 0;
    if (fabs(x.y) > fabs(get(x, d))) d = 1;
    if (fabs(x.z) > fabs(get(x, d))) d = 2;
    const int e = (d + 1) % 3;
    V3 ev = v3(e == 0 ? 1.0 : 0.0, e == 1 ? 1.0 : 0.0, e == 2 ? 1.0 : 0.0);
    b1 = normalized(cross(x, ev));
    b2 = normalized(cross(x, b1));
}

// frame state accessors: q(x,y,z,w) p v bg ba
struct FState {
    Q4 q;
    V3 p, v, bg, ba;
};
XD FState load_state(const double *s) {
    FState f;
    f.q = Q4{s[0], s[1], s[2], s[3]};
    f.p = v3(s[4], s[5], s[6]);
    f.v = v3(s[7], s[8], s[9]);
    f.bg = v3(s[10], s[11], s[12]);
    f.ba = v3(s[13], s[14], s[15]);
    return f;
}
struct Ext {
    Q4 q;
    V3 p;
};

// 2x3 row-major helper: out = A(2x3) * B(3x3)
XD void mul23(const double *A, const M3 &B, double *out) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) out[3 * i + j] = A[3 * i] * B.m[j] + A[3 * i + 1] * B.m[3 + j] + A[3 * i + 2] * B.m[6 + j];
}

// CeresReprojectionErrorFactor::Evaluate.  Jt, Jr: 2x6 row-major (q local 3 | p 3); Jl: 2.
// Outputs are NOT robustified.  want_j == false skips the Jacobians.
XD void eval_reprojection(const FState &tgt, const FState &ref, double inv_depth, V3 z_tgt, V3 z_ref, const Ext &cam,
                          double sx, double sy, double *r, bool want_j, double *Jt, double *Jr, double *Jl) {
    V3 b1, b2;
    s2_tangential_basis(z_tgt, b1, b2);
    V3 y_ref = z_ref / inv_depth;
    V3 y_ref_center = q_rot(cam.q, y_ref) + cam.p;
    V3 x = q_rot(ref.q, y_ref_center) + ref.p;
    V3 y_tgt_center = q_rot(q_conj(tgt.q), x - tgt.p);
    V3 y_tgt = q_rot(q_conj(cam.q), y_tgt_center - cam.p);
    double u0 = dot(b1, y_tgt), u1 = dot(b2, y_tgt), u2 = dot(z_tgt, y_tgt);
    r[0] = sx * (u0 / u2);
    r[1] = sy * (u1 / u2);
    if (!want_j) return;
    // dr_dy_tgt = S * dproj(u) * T^T   (2x3)
    double d00 = 1.0 / u2, d02 = -u0 / (u2 * u2), d11 = 1.0 / u2, d12 = -u1 / (u2 * u2);
    double A[6];
    A[0] = sx * (d00 * b1.x + d02 * z_tgt.x);
    A[1] = sx * (d00 * b1.y + d02 * z_tgt.y);
    A[2] = sx * (d00 * b1.z + d02 * z_tgt.z);
    A[3] = sy * (d11 * b2.x + d12 * z_tgt.x);
    A[4] = sy * (d11 * b2.y + d12 * z_tgt.y);
    A[5] = sy * (d11 * b2.z + d12 * z_tgt.z);
    double B[6], C[6], D[6], E[6];
    mul23(A, q_mat(q_conj(cam.q)), B);    // dr_dy_tgt_center
    mul23(B, q_mat(q_conj(tgt.q)), C);    // dr_dx
    mul23(C, q_mat(ref.q), D);            // dr_dy_ref_center
    mul23(B, hat(y_tgt_center), E);       // dr_dq_tgt
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jt[6 * i + j] = E[3 * i + j];
            Jt[6 * i + 3 + j] = -C[3 * i + j];
            Jr[6 * i + 3 + j] = C[3 * i + j];
        }
    mul23(D, hat(y_ref_center), E);       // -dr_dq_ref
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Jr[6 * i + j] = -E[3 * i + j];
    V3 t = q_mat(cam.q) * y_ref;
    Jl[0] = -(D[0] * t.x + D[1] * t.y + D[2] * t.z) / inv_depth;
    Jl[1] = -(D[3] * t.x + D[4] * t.y + D[5] * t.z) / inv_depth;
}

// The same factor for a landmark that is held CONSTANT (every landmark of localize_newframe / refine_subwindow / PnP), cut in two:
// what depends only on the measurement, the landmark and -- when it is constant too -- the reference frame is evaluated once per
// solve (reprojection_constants), the rest once per evaluation (eval_reprojection_cached).  Expression for expression the body of
// eval_reprojection (same operands, same order): r, Jt, Jr are bit-identical; Jl is not produced (the landmark has no column).
struct ObsConst {
    V3 b1, b2;            // tangent basis at z_tgt
    V3 y_ref_center;      // landmark in the reference camera rig's body frame
    V3 x;                 // landmark in the world (valid when the reference frame is constant)
};
XD ObsConst reprojection_constants(const FState &ref, double inv_depth, V3 z_tgt, V3 z_ref, const Ext &cam) {
    ObsConst c;
    s2_tangential_basis(z_tgt, c.b1, c.b2);
    V3 y_ref = z_ref / inv_depth;
    c.y_ref_center = q_rot(cam.q, y_ref) + cam.p;
    c.x = q_rot(ref.q, c.y_ref_center) + ref.p;
    return c;
}
// ref_free: the reference frame moves (x is recomputed, Jr is produced); otherwise Jr is left untouched.
XD void eval_reprojection_cached(const FState &tgt, const FState &ref, bool ref_free, const ObsConst &c, V3 z_tgt, const Ext &cam,
                                 double sx, double sy, double *r, bool want_j, double *Jt, double *Jr) {
    const V3 b1 = c.b1, b2 = c.b2, y_ref_center = c.y_ref_center;
    V3 x = ref_free ? q_rot(ref.q, y_ref_center) + ref.p : c.x;
    V3 y_tgt_center = q_rot(q_conj(tgt.q), x - tgt.p);
    V3 y_tgt = q_rot(q_conj(cam.q), y_tgt_center - cam.p);
    double u0 = dot(b1, y_tgt), u1 = dot(b2, y_tgt), u2 = dot(z_tgt, y_tgt);
    r[0] = sx * (u0 / u2);
    r[1] = sy * (u1 / u2);
    if (!want_j) return;
    double d00 = 1.0 / u2, d02 = -u0 / (u2 * u2), d11 = 1.0 / u2, d12 = -u1 / (u2 * u2);
    double A[6];
    A[0] = sx * (d00 * b1.x + d02 * z_tgt.x);
    A[1] = sx * (d00 * b1.y + d02 * z_tgt.y);
    A[2] = sx * (d00 * b1.z + d02 * z_tgt.z);
    A[3] = sy * (d11 * b2.x + d12 * z_tgt.x);
    A[4] = sy * (d11 * b2.y + d12 * z_tgt.y);
    A[5] = sy * (d11 * b2.z + d12 * z_tgt.z);
    double B[6], C[6], E[6];
    mul23(A, q_mat(q_conj(cam.q)), B);    // dr_dy_tgt_center
    mul23(B, q_mat(q_conj(tgt.q)), C);    // dr_dx
    mul23(B, hat(y_tgt_center), E);       // dr_dq_tgt
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jt[6 * i + j] = E[3 * i + j];
            Jt[6 * i + 3 + j] = -C[3 * i + j];
        }
    if (!ref_free) return;
    double D[6];
    mul23(C, q_mat(ref.q), D);            // dr_dy_ref_center
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Jr[6 * i + 3 + j] = C[3 * i + j];
    mul23(D, hat(y_ref_center), E);       // -dr_dq_ref
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Jr[6 * i + j] = -E[3 * i + j];
}

// The same factor once more for the case these LDS-resident solves meet nearly always -- constant landmark, constant reference frame,
// free TARGET frame -- with everything that is the same for all factors of one target frame taken out of the factor (round 6): the
// frame's R^T = q_mat(q_conj(q)) and p (frame_table, once per frame and evaluation point) and the camera extrinsics' R_c^T, p_c
// (ext_table, once per solve).  Per factor that leaves two 3x3 matrix-vector products, three dot products and three 2x3 by 3x3
// products: ~180 double-precision instructions where eval_reprojection_cached issues ~450 (quaternion rotations and two q_mat per
// factor), and these kernels are bound by the instructions a wavefront issues.  Same mathematics, another order of operations than
// eval_reprojection_cached: results agree to rounding (tests/test_ba_math_host.py), not bit for bit.
XD void frame_table(const FState &f, double *tab12) {
    const M3 Rt = q_mat(q_conj(f.q));
#pragma unroll
    for (int i = 0; i < 9; ++i) tab12[i] = Rt.m[i];
    tab12[9] = f.p.x;
    tab12[10] = f.p.y;
    tab12[11] = f.p.z;
}
XD void ext_table(const Ext &cam, double *tab12) {
    const M3 Rc = q_mat(q_conj(cam.q));
#pragma unroll
    for (int i = 0; i < 9; ++i) tab12[i] = Rc.m[i];
    tab12[9] = cam.p.x;
    tab12[10] = cam.p.y;
    tab12[11] = cam.p.z;
}
XD void eval_reprojection_tgt(const double *ftab, const double *ctab, const ObsConst &c, V3 z_tgt, double sx, double sy, double *r,
                              bool want_j, double *Jt) {
    M3 RtT, RcT;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        RtT.m[i] = ftab[i];
        RcT.m[i] = ctab[i];
    }
    const V3 b1 = c.b1, b2 = c.b2;
    const V3 y_tgt_center = RtT * (c.x - v3(ftab[9], ftab[10], ftab[11]));
    const V3 y_tgt = RcT * (y_tgt_center - v3(ctab[9], ctab[10], ctab[11]));
    const double u0 = dot(b1, y_tgt), u1 = dot(b2, y_tgt), u2 = dot(z_tgt, y_tgt);
    const double iu2 = 1.0 / u2;
    r[0] = sx * (u0 * iu2);
    r[1] = sy * (u1 * iu2);
    if (!want_j) return;
    const double d02 = -(u0 * iu2) * iu2, d12 = -(u1 * iu2) * iu2;
    double A[6];
    A[0] = sx * (iu2 * b1.x + d02 * z_tgt.x);
    A[1] = sx * (iu2 * b1.y + d02 * z_tgt.y);
    A[2] = sx * (iu2 * b1.z + d02 * z_tgt.z);
    A[3] = sy * (iu2 * b2.x + d12 * z_tgt.x);
    A[4] = sy * (iu2 * b2.y + d12 * z_tgt.y);
    A[5] = sy * (iu2 * b2.z + d12 * z_tgt.z);
    double B[6], C[6];
    mul23(A, RcT, B);    // dr_dy_tgt_center
    mul23(B, RtT, C);    // dr_dx
    const V3 y = y_tgt_center;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // dr_dq_tgt = B hat(y_tgt_center)
        Jt[6 * i + 0] = B[3 * i + 1] * y.z - B[3 * i + 2] * y.y;
        Jt[6 * i + 1] = B[3 * i + 2] * y.x - B[3 * i + 0] * y.z;
        Jt[6 * i + 2] = B[3 * i + 0] * y.y - B[3 * i + 1] * y.x;
        Jt[6 * i + 3] = -C[3 * i + 0];
        Jt[6 * i + 4] = -C[3 * i + 1];
        Jt[6 * i + 5] = -C[3 * i + 2];
    }
}

// CeresRotationPriorFactor::Evaluate.  Jq: 2x3 row-major.
XD void eval_rotation(const FState &tgt, const FState &ref, V3 z_tgt, V3 z_ref, const Ext &cam, double sx, double sy,
                      double *r, bool want_j, double *Jq) {
    V3 b1, b2;
    s2_tangential_basis(z_tgt, b1, b2);
    V3 z_ref_center = q_rot(cam.q, z_ref) + cam.p;
    V3 z_tgt_center = q_rot(q_mul(q_conj(tgt.q), ref.q), z_ref_center);
    V3 z_t = q_rot(q_conj(cam.q), z_tgt_center - cam.p);
    double u0 = dot(b1, z_t), u1 = dot(b2, z_t), u2 = dot(z_tgt, z_t);
    r[0] = sx * (u0 / u2);
    r[1] = sy * (u1 / u2);
    if (!want_j) return;
    double d00 = 1.0 / u2, d02 = -u0 / (u2 * u2), d11 = 1.0 / u2, d12 = -u1 / (u2 * u2);
    double A[6], B[6];
    A[0] = sx * (d00 * b1.x + d02 * z_tgt.x);
    A[1] = sx * (d00 * b1.y + d02 * z_tgt.y);
    A[2] = sx * (d00 * b1.z + d02 * z_tgt.z);
    A[3] = sy * (d11 * b2.x + d12 * z_tgt.x);
    A[4] = sy * (d11 * b2.y + d12 * z_tgt.y);
    A[5] = sy * (d11 * b2.z + d12 * z_tgt.z);
    mul23(A, q_mat(q_conj(cam.q)), B);
    mul23(B, hat(z_tgt_center), Jq);
}

// unpacked view of one XRHIP_IMU_DIM record
struct ImuRec {
    double dt;
    Q4 dq;
    V3 dp, dv;
    M3 dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba;
};
XD ImuRec load_imu(const double *d) {
    ImuRec r;
    r.dt = d[0];
    r.dq = Q4{d[1], d[2], d[3], d[4]};
    r.dp = v3(d[5], d[6], d[7]);
    r.dv = v3(d[8], d[9], d[10]);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.dq_dbg.m[i] = d[11 + i];
        r.dp_dbg.m[i] = d[20 + i];
        r.dp_dba.m[i] = d[29 + i];
        r.dv_dbg.m[i] = d[38 + i];
        r.dv_dba.m[i] = d[47 + i];
    }
    return r;
}

// unwhitened 15-vector residual of CeresPreIntegrationErrorFactor (before sqrt_inv_cov)
XD void imu_raw_residual(const FState &fi, const FState &fj, const ImuRec &pre, V3 bg0, V3 ba0, const Ext &imu,
                         double *r15) {
    const V3 gravity = v3(0.0, 0.0, -9.80665);
    const Q4 q_i = q_mul(fi.q, imu.q);
    const V3 p_i = fi.p + q_rot(fi.q, imu.p);
    const Q4 q_j = q_mul(fj.q, imu.q);
    const V3 p_j = fj.p + q_rot(fj.q, imu.p);
    const double dt = pre.dt;
    const V3 dbg = fi.bg - bg0, dba = fi.ba - ba0;
    V3 rq = logmap(q_mul(q_mul(q_conj(q_mul(pre.dq, expmap(pre.dq_dbg * dbg))), q_conj(q_i)), q_j));
    V3 rp = q_rot(q_conj(q_i), p_j - p_i - fi.v * dt - gravity * (0.5 * dt * dt)) -
            (pre.dp + pre.dp_dbg * dbg + pre.dp_dba * dba);
    V3 rv = q_rot(q_conj(q_i), fj.v - fi.v - gravity * dt) - (pre.dv + pre.dv_dbg * dbg + pre.dv_dba * dba);
    V3 rbg = fj.bg - fi.bg, rba = fj.ba - fi.ba;
    r15[0] = rq.x; r15[1] = rq.y; r15[2] = rq.z;
    r15[3] = rp.x; r15[4] = rp.y; r15[5] = rp.z;
    r15[6] = rv.x; r15[7] = rv.y; r15[8] = rv.z;
    r15[9] = rbg.x; r15[10] = rbg.y; r15[11] = rbg.z;
    r15[12] = rba.x; r15[13] = rba.y; r15[14] = rba.z;
}

// Unwhitened Jacobians of the IMU residual as sparse 3x3 blocks written into dense 15x15 row-major
// arrays (zero-filled by the caller).  rq = the rotation part of the raw residual.
XD void put33(double *J, int r0, int c0, const M3 &b) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) J[15 * (r0 + i) + c0 + j] = b.m[3 * i + j];
}
// need_i / need_j: a side whose frame is constant is left untouched (the caller zeroes it)
XD void imu_raw_jacobians(const FState &fi, const FState &fj, const ImuRec &pre, V3 bg0, V3 ba0, const Ext &imu,
                          V3 rq, double *Ji, double *Jj, bool need_i = true, bool need_j = true) {
    const V3 gravity = v3(0.0, 0.0, -9.80665);
    const Q4 q_i = q_mul(fi.q, imu.q);
    const Q4 q_j = q_mul(fj.q, imu.q);
    const V3 p_j = fj.p + q_rot(fj.q, imu.p);
    const double dt = pre.dt;
    const V3 dbg = fi.bg - bg0;
    const M3 Jr_inv = inverse3(right_jacobian(rq));
    const M3 Rqi_t = q_mat(q_conj(q_i));
    const M3 Rimu_t = q_mat(q_conj(imu.q));
    const M3 I3 = m3_identity();
    if (need_j) {
        put33(Jj, 0, 0, Jr_inv * Rimu_t);
        put33(Jj, 3, 0, -(Rqi_t * q_mat(fj.q) * hat(imu.p)));
        put33(Jj, 3, 3, Rqi_t);
        put33(Jj, 6, 6, Rqi_t);
        put33(Jj, 9, 9, I3);
        put33(Jj, 12, 12, I3);
    }
    if (!need_i) return;
    put33(Ji, 0, 0, -(Jr_inv * q_mat(q_conj(q_j)) * q_mat(fi.q)));
    put33(Ji, 3, 0, Rimu_t * hat(q_rot(q_conj(fi.q), p_j - fi.p - fi.v * dt - gravity * (0.5 * dt * dt))));
    put33(Ji, 6, 0, Rimu_t * hat(q_rot(q_conj(fi.q), fj.v - fi.v - gravity * dt)));
    put33(Ji, 3, 3, -Rqi_t);
    put33(Ji, 3, 6, Rqi_t * (-dt));
    put33(Ji, 6, 6, -Rqi_t);
    put33(Ji, 0, 9, -(Jr_inv * q_mat(q_conj(expmap(rq))) * right_jacobian(pre.dq_dbg * dbg) * pre.dq_dbg));
    put33(Ji, 3, 9, -pre.dp_dbg);
    put33(Ji, 6, 9, -pre.dv_dbg);
    put33(Ji, 9, 9, -I3);
    put33(Ji, 3, 12, -pre.dp_dba);
    put33(Ji, 6, 12, -pre.dv_dba);
    put33(Ji, 12, 12, -I3);
}

// The same Jacobians in four independent parts (part = 0..3), for four wavefronts working on ONE factor: every block
// of imu_raw_jacobians is produced by exactly one part, with the same expression -- the union of the four calls writes
// what the single call writes.  Part 0 and 1 carry the two long chains (right Jacobian + its inverse; expmap + a second
// right Jacobian + three products), part 2 and 3 the rotation blocks.
XD void imu_raw_jacobians_part(int part, const FState &fi, const FState &fj, const ImuRec &pre, V3 bg0, V3 ba0, const Ext &imu,
                               V3 rq, double *Ji, double *Jj, bool need_i, bool need_j) {
    const V3 gravity = v3(0.0, 0.0, -9.80665);
    const double dt = pre.dt;
    const M3 I3 = m3_identity();
    if (part == 0) {
        const M3 Jr_inv = inverse3(right_jacobian(rq));
        if (need_j) put33(Jj, 0, 0, Jr_inv * q_mat(q_conj(imu.q)));
        if (need_i) {
            const Q4 q_j = q_mul(fj.q, imu.q);
            put33(Ji, 0, 0, -(Jr_inv * q_mat(q_conj(q_j)) * q_mat(fi.q)));
        }
    } else if (part == 1) {
        if (need_i) {
            const V3 dbg = fi.bg - bg0;
            const M3 Jr_inv = inverse3(right_jacobian(rq));
            put33(Ji, 0, 9, -(Jr_inv * q_mat(q_conj(expmap(rq))) * right_jacobian(pre.dq_dbg * dbg) * pre.dq_dbg));
        }
    } else if (part == 2) {
        if (need_i) {
            const M3 Rimu_t = q_mat(q_conj(imu.q));
            const V3 p_j = fj.p + q_rot(fj.q, imu.p);
            put33(Ji, 3, 0, Rimu_t * hat(q_rot(q_conj(fi.q), p_j - fi.p - fi.v * dt - gravity * (0.5 * dt * dt))));
            put33(Ji, 6, 0, Rimu_t * hat(q_rot(q_conj(fi.q), fj.v - fi.v - gravity * dt)));
            put33(Ji, 3, 9, -pre.dp_dbg);
            put33(Ji, 6, 9, -pre.dv_dbg);
            put33(Ji, 9, 9, -I3);
            put33(Ji, 3, 12, -pre.dp_dba);
            put33(Ji, 6, 12, -pre.dv_dba);
            put33(Ji, 12, 12, -I3);
        }
    } else {
        const Q4 q_i = q_mul(fi.q, imu.q);
        const M3 Rqi_t = q_mat(q_conj(q_i));
        if (need_j) {
            put33(Jj, 3, 0, -(Rqi_t * q_mat(fj.q) * hat(imu.p)));
            put33(Jj, 3, 3, Rqi_t);
            put33(Jj, 6, 6, Rqi_t);
            put33(Jj, 9, 9, I3);
            put33(Jj, 12, 12, I3);
        }
        if (need_i) {
            put33(Ji, 3, 3, -Rqi_t);
            put33(Ji, 3, 6, Rqi_t * (-dt));
            put33(Ji, 6, 6, -Rqi_t);
        }
    }
}

// ---- the same residual and Jacobians cut along their data dependencies, for SEVERAL wavefronts working on one factor
// (ba_chain.hip.h).  A double-precision instruction costs its wavefront ~8 cycles of issue whatever the number of active lanes,
// so one lane's stream of ~2200 instructions is 8 us; the only way down is more streams.  Every value below is produced by the
// expression imu_raw_residual / imu_raw_jacobians use for it, operand for operand and in the same order (left-associated
// products included): the pieces put together are bit-identical to the single-lane forms.
//   rq chain (long: expmap, three quaternion products, logmap)          imu_residual_rq
//   everything else of the residual (no transcendental)                imu_residual_rest -> r15[3..15)
//   matrices that do not depend on rq                                   imu_jac_pre
//   Jr^-1(rq) | R(expmap(rq))^T                                         imu_jac_jrinv | imu_jac_B   (need rq)
//   the products                                                        imu_jac_finish0 | imu_jac_finish1
XD V3 imu_residual_rq(const FState &fi, const FState &fj, const ImuRec &pre, V3 bg0, const Ext &imu) {
    const Q4 q_i = q_mul(fi.q, imu.q);
    const Q4 q_j = q_mul(fj.q, imu.q);
    const V3 dbg = fi.bg - bg0;
    return logmap(q_mul(q_mul(q_conj(q_mul(pre.dq, expmap(pre.dq_dbg * dbg))), q_conj(q_i)), q_j));
}
XD void imu_residual_rest(const FState &fi, const FState &fj, const ImuRec &pre, V3 bg0, V3 ba0, const Ext &imu, double *r15) {
    const V3 gravity = v3(0.0, 0.0, -9.80665);
    const Q4 q_i = q_mul(fi.q, imu.q);
    const V3 p_i = fi.p + q_rot(fi.q, imu.p);
    const V3 p_j = fj.p + q_rot(fj.q, imu.p);
    const double dt = pre.dt;
    const V3 dbg = fi.bg - bg0, dba = fi.ba - ba0;
    V3 rp = q_rot(q_conj(q_i), p_j - p_i - fi.v * dt - gravity * (0.5 * dt * dt)) -
            (pre.dp + pre.dp_dbg * dbg + pre.dp_dba * dba);
    V3 rv = q_rot(q_conj(q_i), fj.v - fi.v - gravity * dt) - (pre.dv + pre.dv_dbg * dbg + pre.dv_dba * dba);
    V3 rbg = fj.bg - fi.bg, rba = fj.ba - fi.ba;
    r15[3] = rp.x; r15[4] = rp.y; r15[5] = rp.z;
    r15[6] = rv.x; r15[7] = rv.y; r15[8] = rv.z;
    r15[9] = rbg.x; r15[10] = rbg.y; r15[11] = rbg.z;
    r15[12] = rba.x; r15[13] = rba.y; r15[14] = rba.z;
}
XD void store33(double *d, const M3 &a) {
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.m[i];
}
XD M3 load33(const double *d) {
    M3 a;
#pragma unroll
    for (int i = 0; i < 9; ++i) a.m[i] = d[i];
    return a;
}
constexpr int IMU_XCH = 54;   // doubles exchanged per factor: RJth, Q1, Q2, Rimu_t (imu_jac_pre) | B | Jr_inv
XD void imu_jac_pre(const FState &fi, const FState &fj, const ImuRec &pre, V3 bg0, const Ext &imu, double *x36, bool need_i = true,
                    bool need_j = true) {
    if (need_i) {   // only the Jacobian with respect to frame i uses these
        const V3 dbg = fi.bg - bg0;
        const Q4 q_j = q_mul(fj.q, imu.q);
        store33(x36, right_jacobian(pre.dq_dbg * dbg));
        store33(x36 + 9, q_mat(q_conj(q_j)));
        store33(x36 + 18, q_mat(fi.q));
    }
    if (need_j) store33(x36 + 27, q_mat(q_conj(imu.q)));
}
XD M3 imu_jac_jrinv(V3 rq) { return inverse3(right_jacobian(rq)); }
XD M3 imu_jac_B(V3 rq) { return q_mat(q_conj(expmap(rq))); }
XD void imu_jac_finish0(const M3 &Jr_inv, const double *x36, double *Ji, double *Jj, bool need_i, bool need_j) {
    if (need_j) put33(Jj, 0, 0, Jr_inv * load33(x36 + 27));
    if (need_i) put33(Ji, 0, 0, -(Jr_inv * load33(x36 + 9) * load33(x36 + 18)));
}
XD void imu_jac_finish1(const M3 &Jr_inv, const M3 &B, const double *x36, const M3 &dq_dbg, double *Ji, bool need_i) {
    if (need_i) put33(Ji, 0, 9, -(Jr_inv * B * load33(x36) * dq_dbg));
}

// QuaternionParameterization::Plus + additive blocks; d15 = (dq3, dp, dv, dbg, dba); mask bit0: pose free, bit1: motion free
XD void state_plus(const double *s, const double *d15, bool pose_free, bool motion_free, double *out) {
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = s[i];
    if (pose_free) {
        Q4 q = q_normalized(q_mul(Q4{s[0], s[1], s[2], s[3]}, expmap(v3(d15[0], d15[1], d15[2]))));
        out[0] = q.x;
        out[1] = q.y;
        out[2] = q.z;
        out[3] = q.w;
        out[4] = s[4] + d15[3];
        out[5] = s[5] + d15[4];
        out[6] = s[6] + d15[5];
    }
    if (motion_free) {
#pragma unroll
        for (int i = 0; i < 9; ++i) out[7 + i] = s[7 + i] + d15[6 + i];
    }
}

}   // namespace xrhip
