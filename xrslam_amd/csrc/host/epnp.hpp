// epnp.hpp -- EPnP pose from n >= 4 point / bearing pairs (Lepetit, Moreno-Noguer, Fua, IJCV 2009), the minimal
// solver of the RD-VIO outlier filter.
//
// The reference obtains it from OpenCV (geometry/pnp.h:10-41: cv::solvePnP(P3D, P2D, I, noArray, rvec, tvec, false,
// CV_EPNP) on 6 points given as float32, result rounded to float32 and passed through cv::Rodrigues).  OpenCV is not
// part of this image: this is a restatement of the published algorithm in the arrangement of OpenCV's epnp.cpp
// (4 control points from the PCA of the object points, null space of M^T M, three beta approximations each refined
// by 5 Gauss-Newton steps, rigid alignment by SVD, best reprojection error wins).  Inputs are rounded to float32 and
// the pose to what the float32 rvec / tvec round trip leaves, like the reference's wrapper.  PARITY UNPINNED against
// OpenCV; checked against ground-truth poses in tests/test_parsac.py.
#pragma once
#include "two_view.hpp"

namespace xrh {

struct Pose34 {   // x_cam = R x_world + t
    M3 R = M3::identity();
    V3 t;
};

namespace epnp_detail {

inline double dist2(const V3 &a, const V3 &b) { return dot(a - b, a - b); }

struct Work {
    int n = 0;
    std::vector<V3> pw;
    std::vector<V2> us;
    V3 cws[4];
    std::vector<std::array<double, 4>> alphas;
    V3 ccs[4];
    std::vector<V3> pcs;
};

inline void choose_control_points(Work &w) {
    V3 c{0, 0, 0};
    for (const V3 &p : w.pw) c = c + p;
    c = c / (double)w.n;
    w.cws[0] = c;
    Dense C(3, 3), V;
    for (const V3 &p : w.pw) {
        const V3 d = p - c;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) C(i, j) += d[i] * d[j];
    }
    std::vector<double> s;
    jacobi_svd(C, s, V);   // symmetric PSD: singular values = eigenvalues (descending), columns of V = eigenvectors
    for (int i = 1; i < 4; ++i) {
        const double k = std::sqrt(s[i - 1] / w.n);
        w.cws[i] = c + V3{V(0, i - 1), V(1, i - 1), V(2, i - 1)} * k;
    }
}

inline void barycentric_coordinates(Work &w) {
    M3 CC;
    for (int i = 0; i < 3; ++i)
        for (int j = 1; j < 4; ++j) CC(i, j - 1) = w.cws[j][i] - w.cws[0][i];
    const M3 Ci = inverse(CC);
    w.alphas.resize(w.n);
    for (int k = 0; k < w.n; ++k) {
        const V3 d = w.pw[k] - w.cws[0];
        const V3 a = Ci * d;
        w.alphas[k] = {1.0 - a.x - a.y - a.z, a.x, a.y, a.z};
    }
}

// rigid alignment of the camera-frame points to the world points (Arun et al.), returns the mean reprojection error
inline double align_and_score(Work &w, const Dense &V12, const double betas[4], Pose34 &pose) {
    for (int i = 0; i < 4; ++i) w.ccs[i] = V3{0, 0, 0};
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 3; ++c) w.ccs[i][c] += betas[k] * V12(3 * i + c, 11 - k);
    w.pcs.resize(w.n);
    for (int k = 0; k < w.n; ++k) {
        V3 p{0, 0, 0};
        for (int j = 0; j < 4; ++j) p = p + w.ccs[j] * w.alphas[k][j];
        w.pcs[k] = p;
    }
    if (w.pcs[0].z < 0.0) {   // the null vectors are defined up to sign: points must lie in front of the camera
        for (int i = 0; i < 4; ++i) w.ccs[i] = -w.ccs[i];
        for (V3 &p : w.pcs) p = -p;
    }
    V3 pc0{0, 0, 0}, pw0{0, 0, 0};
    for (int k = 0; k < w.n; ++k) {
        pc0 = pc0 + w.pcs[k];
        pw0 = pw0 + w.pw[k];
    }
    pc0 = pc0 / (double)w.n;
    pw0 = pw0 / (double)w.n;
    Dense ABt(3, 3), V, U;
    for (int k = 0; k < w.n; ++k) {
        const V3 a = w.pcs[k] - pc0, b = w.pw[k] - pw0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) ABt(i, j) += a[i] * b[j];
    }
    std::vector<double> s;
    jacobi_svd(ABt, s, V, &U);
    // complete U to an orthonormal basis if the configuration is (numerically) planar
    V3 u0{U(0, 0), U(1, 0), U(2, 0)}, u1{U(0, 1), U(1, 1), U(2, 1)}, u2{U(0, 2), U(1, 2), U(2, 2)};
    if (!(s[2] > 1e-12 * s[0])) {
        u0 = normalized(u0);
        u1 = normalized(u1 - u0 * dot(u0, u1));
        u2 = cross(u0, u1);
    }
    M3 Um, Vt;
    for (int r = 0; r < 3; ++r) {
        Um(r, 0) = u0[r];
        Um(r, 1) = u1[r];
        Um(r, 2) = u2[r];
        for (int c = 0; c < 3; ++c) Vt(r, c) = V(c, r);
    }
    M3 R = Um * Vt;
    if (det(R) < 0)
        for (int c = 0; c < 3; ++c) R(2, c) = -R(2, c);
    pose.R = R;
    pose.t = pc0 - R * pw0;
    double err = 0;
    for (int k = 0; k < w.n; ++k) {
        const V3 q = R * w.pw[k] + pose.t;
        const double du = w.us[k].x - q.x / q.z, dv = w.us[k].y - q.y / q.z;
        err += std::sqrt(du * du + dv * dv);
    }
    return err / w.n;
}

inline void gauss_newton(const double L[6][10], const double rho[6], double b[4]) {
    for (int it = 0; it < 5; ++it) {
        Dense A(6, 4);
        std::vector<double> r(6);
        for (int i = 0; i < 6; ++i) {
            const double *l = L[i];
            A(i, 0) = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
            A(i, 1) = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
            A(i, 2) = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
            A(i, 3) = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
            r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                             l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
        }
        const std::vector<double> x = lstsq_qr(A, r);
        for (int k = 0; k < 4; ++k) b[k] += x[k];
    }
}

inline std::vector<double> solve_columns(const double L[6][10], const double rho[6], std::initializer_list<int> cols) {
    Dense A(6, (int)cols.size());
    int c = 0;
    for (int col : cols) {
        for (int i = 0; i < 6; ++i) A(i, c) = L[i][col];
        ++c;
    }
    return lstsq_qr(A, std::vector<double>(rho, rho + 6));
}

inline double f32(double v) { return (double)(float)v; }

}   // namespace epnp_detail

inline Pose34 solve_pnp_epnp(const V3 *Xs, const V2 *xs, int n) {
    using namespace epnp_detail;
    Work w;
    w.n = n;
    for (int i = 0; i < n; ++i) {   // the reference hands cv::Point3f / cv::Point2f to OpenCV
        w.pw.push_back({f32(Xs[i].x), f32(Xs[i].y), f32(Xs[i].z)});
        w.us.push_back({f32(xs[i].x), f32(xs[i].y)});
    }
    choose_control_points(w);
    barycentric_coordinates(w);
    Dense M(2 * n, 12);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 4; ++j) {
            const double a = w.alphas[i][j];
            M(2 * i, 3 * j) = a;
            M(2 * i, 3 * j + 2) = -a * w.us[i].x;
            M(2 * i + 1, 3 * j + 1) = a;
            M(2 * i + 1, 3 * j + 2) = -a * w.us[i].y;
        }
    Dense MtM(12, 12), V12;
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) {
            double s = 0;
            for (int k = 0; k < 2 * n; ++k) s += M(k, i) * M(k, j);
            MtM(i, j) = s;
        }
    std::vector<double> ev;
    jacobi_svd(MtM, ev, V12);   // columns 11, 10, 9, 8: the four smallest eigenvalues
    // distances between control points as quadratic forms in the betas: B11 B12 B22 B13 B23 B33 B14 B24 B34 B44
    static const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
    double L[6][10], rho[6];
    for (int r = 0; r < 6; ++r) {
        V3 dv[4];
        for (int k = 0; k < 4; ++k)
            for (int c = 0; c < 3; ++c) dv[k][c] = V12(3 * PA[r] + c, 11 - k) - V12(3 * PB[r] + c, 11 - k);
        L[r][0] = dot(dv[0], dv[0]);
        L[r][1] = 2 * dot(dv[0], dv[1]);
        L[r][2] = dot(dv[1], dv[1]);
        L[r][3] = 2 * dot(dv[0], dv[2]);
        L[r][4] = 2 * dot(dv[1], dv[2]);
        L[r][5] = dot(dv[2], dv[2]);
        L[r][6] = 2 * dot(dv[0], dv[3]);
        L[r][7] = 2 * dot(dv[1], dv[3]);
        L[r][8] = 2 * dot(dv[2], dv[3]);
        L[r][9] = dot(dv[3], dv[3]);
        rho[r] = dist2(w.cws[PA[r]], w.cws[PB[r]]);
    }
    double betas[3][4];
    {   // N = 4 unknowns linearised: B11 B12 B13 B14
        const std::vector<double> b4 = solve_columns(L, rho, {0, 1, 3, 6});
        double *b = betas[0];
        if (b4[0] < 0) {
            b[0] = std::sqrt(-b4[0]);
            b[1] = -b4[1] / b[0];
            b[2] = -b4[2] / b[0];
            b[3] = -b4[3] / b[0];
        } else {
            b[0] = std::sqrt(b4[0]);
            b[1] = b4[1] / b[0];
            b[2] = b4[2] / b[0];
            b[3] = b4[3] / b[0];
        }
    }
    {   // N = 2: B11 B12 B22
        const std::vector<double> b3 = solve_columns(L, rho, {0, 1, 2});
        double *b = betas[1];
        if (b3[0] < 0) {
            b[0] = std::sqrt(-b3[0]);
            b[1] = b3[2] < 0 ? std::sqrt(-b3[2]) : 0.0;
        } else {
            b[0] = std::sqrt(b3[0]);
            b[1] = b3[2] > 0 ? std::sqrt(b3[2]) : 0.0;
        }
        if (b3[1] < 0) b[0] = -b[0];
        b[2] = b[3] = 0.0;
    }
    {   // N = 3: B11 B12 B22 B13 B23
        const std::vector<double> b5 = solve_columns(L, rho, {0, 1, 2, 3, 4});
        double *b = betas[2];
        if (b5[0] < 0) {
            b[0] = std::sqrt(-b5[0]);
            b[1] = b5[2] < 0 ? std::sqrt(-b5[2]) : 0.0;
        } else {
            b[0] = std::sqrt(b5[0]);
            b[1] = b5[2] > 0 ? std::sqrt(b5[2]) : 0.0;
        }
        if (b5[1] < 0) b[0] = -b[0];
        b[2] = b5[3] / b[0];
        b[3] = 0.0;
    }
    Pose34 best;
    double best_err = std::numeric_limits<double>::infinity();
    for (int k = 0; k < 3; ++k) {
        gauss_newton(L, rho, betas[k]);
        Pose34 p;
        const double e = align_and_score(w, V12, betas[k], p);
        if (e < best_err) {   // NaN never wins
            best_err = e;
            best = p;
        }
    }
    // rvec / tvec leave the reference's wrapper as float32 and the rotation is rebuilt from the rounded rvec
    const V3 rv = logmap(quat_from_matrix(best.R));
    const V3 rf{f32(rv.x), f32(rv.y), f32(rv.z)};
    M3 R = to_matrix(expmap(rf));
    for (double &v : R.m) v = f32(v);
    Pose34 out;
    out.R = R;
    out.t = {f32(best.t.x), f32(best.t.y), f32(best.t.z)};
    return out;
}

}   // namespace xrh
