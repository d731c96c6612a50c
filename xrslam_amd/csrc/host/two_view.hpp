// two_view.hpp -- two-view relative pose pieces and the small dense solves used by the initialiser.
//
// Host-side mirror of (file:line under /root/reference/xrslam/src/xrslam):
//   solve_homography_4pt / decompose_homography   geometry/homography.cpp:5-158 (Malis-Vargas analytic decomposition)
//   homography_geometric_error                    geometry/homography.h:17-21
//   find_homography_matrix                        geometry/stereo.cpp:93-118
//   decompose_essential (SVD branch)              geometry/essential.cpp:270-283
//   triangulate_point (two views)                 geometry/stereo.h:72-82
//   s2_tangential_basis                           geometry/lie_algebra.cpp:47-56
//   logmap                                        geometry/lie_algebra.h:20-23 (Eigen::AngleAxisd(q))
// and of the Eigen calls the initialiser makes: Quaterniond(Matrix3d), Quaterniond::FromTwoVectors,
// Matrix3d::inverse, JacobiSVD<Matrix3d>::solve, FullPivHouseholderQR::solve (least squares).
//
// Runs once per initialisation attempt on <= a few hundred correspondences; sequential on the host like the
// reference.  PARITY UNPINNED against the reference binary (needs Eigen, which is not in this image): checked
// against numpy / ground-truth geometry in tests/test_two_view.py.
#pragma once
#include "geometry.hpp"

namespace xrh {

inline M3 inverse(const M3 &a) {   // cofactor form (what Eigen uses for 3x3)
    M3 r;
    r(0, 0) = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
    r(0, 1) = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
    r(0, 2) = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
    r(1, 0) = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
    r(1, 1) = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
    r(1, 2) = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
    r(2, 0) = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
    r(2, 1) = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
    r(2, 2) = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
    const double d = a(0, 0) * r(0, 0) + a(0, 1) * r(1, 0) + a(0, 2) * r(2, 0);
    for (double &v : r.m) v /= d;
    return r;
}
inline M3 scaled(const M3 &a, double s) {
    M3 r = a;
    for (double &v : r.m) v *= s;
    return r;
}
inline V3 col(const M3 &a, int c) { return {a(0, c), a(1, c), a(2, c)}; }
inline M3 outer(V3 a, V3 b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = a[i] * b[j];
    return r;
}
inline M3 operator-(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i];
    return r;
}

// rotation matrix -> unit quaternion, branch on the trace / largest diagonal element (Eigen's quaternionbase_assign)
inline Quat quat_from_matrix(const M3 &R) {
    Quat q;
    double t = R(0, 0) + R(1, 1) + R(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R(2, 1) - R(1, 2)) * t;
        q.y = (R(0, 2) - R(2, 0)) * t;
        q.z = (R(1, 0) - R(0, 1)) * t;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R(k, j) - R(j, k)) * t;
        v[j] = (R(j, i) + R(i, j)) * t;
        v[k] = (R(k, i) + R(i, k)) * t;
        q.x = v[0];
        q.y = v[1];
        q.z = v[2];
    }
    return q;
}

inline V3 logmap(Quat q) {   // angle * axis, angle in [0, pi]
    V3 v{q.x, q.y, q.z};
    double n = norm(v);
    if (n == 0.0) return {0, 0, 0};
    double angle = 2.0 * std::atan2(n, std::fabs(q.w));
    if (q.w < 0) n = -n;
    return v * (angle / n);
}

// shortest rotation taking the direction of a to the direction of b
inline Quat quat_from_two_vectors(V3 a, V3 b) {
    V3 v0 = normalized(a), v1 = normalized(b);
    double c = dot(v1, v0);
    if (c < -1.0 + 1e-12) {   // opposite: any axis perpendicular to v0, half turn
        V3 e = std::fabs(v0.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
        V3 axis = normalized(cross(v0, e));
        double w2 = (1.0 + c) * 0.5;
        double s = std::sqrt(std::max(0.0, 1.0 - w2));
        return {axis.x * s, axis.y * s, axis.z * s, std::sqrt(std::max(0.0, w2))};
    }
    V3 axis = cross(v0, v1);
    double s = std::sqrt((1.0 + c) * 2.0), inv = 1.0 / s;
    return {axis.x * inv, axis.y * inv, axis.z * inv, s * 0.5};
}

inline void s2_tangential_basis(V3 x, V3 &b1, V3 &b2) {
    int d = 0;
    for (int i = 1; i < 3; ++i)
        if (std::fabs(x[i]) > std::fabs(x[d])) d = i;
    V3 e{0, 0, 0};
    e[(d + 1) % 3] = 1.0;
    b1 = normalized(cross(x, e));
    b2 = normalized(cross(x, b1));
}

// ------------------------------------------------------------------------------------------ homography
// p2 ~ H p1 from four correspondences: isotropic normalisation (centroid, mean distance sqrt 2), DLT null vector.
inline M3 solve_homography_4pt(const std::array<V2, 4> &pa, const std::array<V2, 4> &pb) {
    V2 ma, mb;
    for (int i = 0; i < 4; ++i) {
        ma.x += pa[i].x; ma.y += pa[i].y;
        mb.x += pb[i].x; mb.y += pb[i].y;
    }
    ma.x /= 4; ma.y /= 4; mb.x /= 4; mb.y /= 4;
    double sa = 0, sb = 0;
    for (int i = 0; i < 4; ++i) {
        sa += std::hypot(pa[i].x - ma.x, pa[i].y - ma.y);
        sb += std::hypot(pb[i].x - mb.x, pb[i].y - mb.y);
    }
    const double r2 = std::sqrt(2.0);
    sa = 1.0 / (r2 * sa);
    sb = 1.0 / (r2 * sb);
    Dense A(8, 9);
    for (int i = 0; i < 4; ++i) {
        const double ax = (pa[i].x - ma.x) * sa, ay = (pa[i].y - ma.y) * sa;
        const double bx = (pb[i].x - mb.x) * sb, by = (pb[i].y - mb.y) * sb;
        // unknown h holds H column by column (h[3c + r] = H(r, c)); the two rows are  b x (H a) = 0
        A(2 * i, 1) = -ax;     A(2 * i, 2) = ax * by;
        A(2 * i, 4) = -ay;     A(2 * i, 5) = ay * by;
        A(2 * i, 7) = -1;      A(2 * i, 8) = by;
        A(2 * i + 1, 0) = ax;  A(2 * i + 1, 2) = -ax * bx;
        A(2 * i + 1, 3) = ay;  A(2 * i + 1, 5) = -ay * bx;
        A(2 * i + 1, 6) = 1;   A(2 * i + 1, 8) = -bx;
    }
    std::vector<double> s;
    Dense V;
    jacobi_svd(A, s, V);
    M3 NH;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) NH(r, c) = V(3 * c + r, 8);
    M3 Na, Nb;
    Nb(0, 0) = 1 / sb; Nb(0, 2) = mb.x; Nb(1, 1) = 1 / sb; Nb(1, 2) = mb.y; Nb(2, 2) = 1;
    Na(0, 0) = sa; Na(0, 2) = -sa * ma.x; Na(1, 1) = sa; Na(1, 2) = -sa * ma.y; Na(2, 2) = 1;
    return Nb * NH * Na;
}

inline double homography_geometric_error(const M3 &H, V2 p1, V2 p2) {
    V3 q = H * V3{p1.x, p1.y, 1.0};
    const double dx = p2.x - q.x / q.z, dy = p2.y - q.y / q.z;
    return dx * dx + dy * dy;
}

inline M3 find_homography_matrix(const std::vector<V2> &p1, const std::vector<V2> &p2, std::vector<char> &mask,
                                 double threshold = 1.0, double confidence = 0.999, size_t max_iteration = 1000,
                                 int seed = 0) {
    mask.clear();
    auto solver = [](const std::array<V2, 4> &a, const std::array<V2, 4> &b) {
        return std::vector<M3>{solve_homography_4pt(a, b)};
    };
    auto make_eval = [](const M3 &H) {
        M3 Hi = inverse(H);
        return [H, Hi](V2 a, V2 b) { return homography_geometric_error(H, a, b) + homography_geometric_error(Hi, b, a); };
    };
    return ransac_solve<4>(p1, p2, 2.0 * 5.99 * threshold * threshold, confidence, max_iteration, seed, solver,
                           make_eval, mask);
}

// H = R (I + t n^T / d) up to scale: the two physically distinct (R, t, n) of the analytic decomposition.
// Returns false for a pure rotation (H^T H = I after scaling by the middle singular value).
// The sign of H is chosen with det > 0 (a plane seen from the same side by both cameras); the reference leaves
// the sign to its SVD and, when that comes out negative, its two homography candidates are improper rotations
// which lose the triangulation vote -- see DESIGN.md "deviations".
inline bool decompose_homography(const M3 &H_in, M3 &R1, M3 &R2, V3 &T1, V3 &T2, V3 &n1, V3 &n2) {
    Dense Hd(3, 3), V;
    for (int i = 0; i < 9; ++i) Hd.a[i] = H_in.m[i];
    std::vector<double> sv;
    jacobi_svd(Hd, sv, V);
    M3 Hn = scaled(H_in, (det(H_in) < 0 ? -1.0 : 1.0) / sv[1]);
    M3 S = transpose(Hn) * Hn - M3::identity();
    bool pure_rotation = true;
    for (double v : S.m)
        if (std::fabs(v) > 1e-3) pure_rotation = false;
    if (pure_rotation) {
        R1 = R2 = Hn;
        T1 = T2 = n1 = n2 = V3{0, 0, 0};
        return false;
    }
    // opposites of the principal minors of S
    const double m00 = S(1, 2) * S(1, 2) - S(1, 1) * S(2, 2);
    const double m11 = S(0, 2) * S(0, 2) - S(0, 0) * S(2, 2);
    const double m22 = S(0, 1) * S(0, 1) - S(0, 0) * S(1, 1);
    const double r00 = std::sqrt(m00), r11 = std::sqrt(m11), r22 = std::sqrt(m22);
    const double tr = S(0, 0) + S(1, 1) + S(2, 2);
    const double nu = 2.0 * std::sqrt(1 + tr - m00 - m11 - m22);
    const double te2 = 2 + tr - nu;
    V3 ts1, ts2;
    auto sgn = [](double v) { return v < 0 ? -1.0 : 1.0; };
    if (S(0, 0) > S(1, 1) && S(0, 0) > S(2, 2)) {
        const double e = sgn(S(0, 1) * S(0, 2) - S(0, 0) * S(1, 2));
        n1 = {S(0, 0), S(0, 1) + r22, S(0, 2) + e * r11};
        n2 = {S(0, 0), S(0, 1) - r22, S(0, 2) - e * r11};
        ts1 = n2 * (norm(n1) / S(0, 0));
        ts2 = n1 * (norm(n2) / S(0, 0));
    } else if (S(1, 1) > S(0, 0) && S(1, 1) > S(2, 2)) {
        const double e = sgn(S(1, 1) * S(0, 2) - S(0, 1) * S(1, 2));
        n1 = {S(0, 1) + r22, S(1, 1), S(1, 2) - e * r00};
        n2 = {S(0, 1) - r22, S(1, 1), S(1, 2) + e * r00};
        ts1 = n2 * (norm(n1) / S(1, 1));
        ts2 = n1 * (norm(n2) / S(1, 1));
    } else {
        const double e = sgn(S(1, 2) * S(0, 2) - S(0, 1) * S(2, 2));
        n1 = {S(0, 2) + e * r11, S(1, 2) + r00, S(2, 2)};
        n2 = {S(0, 2) - e * r11, S(1, 2) - r00, S(2, 2)};
        ts1 = n2 * (norm(n1) / S(2, 2));
        ts2 = n1 * (norm(n2) / S(2, 2));
    }
    n1 = normalized(n1);
    n2 = normalized(n2);
    ts1 = ts1 - n1 * te2;
    ts2 = ts2 - n2 * te2;
    R1 = Hn * (M3::identity() - outer(ts1 / nu, n1));
    R2 = Hn * (M3::identity() - outer(ts2 / nu, n2));
    T1 = R1 * (ts1 * 0.5);
    T2 = R2 * (ts2 * 0.5);
    return true;
}

// ------------------------------------------------------------------------------------------- essential
// E = U diag(1,1,0) V^T -> the twisted pair R1 = U W V^T, R2 = U W^T V^T and the baseline direction u3.
inline void decompose_essential(const M3 &E, M3 &R1, M3 &R2, V3 &T) {
    Dense Ed(3, 3), Vd, Ud;
    for (int i = 0; i < 9; ++i) Ed.a[i] = E.m[i];
    std::vector<double> sv;
    jacobi_svd(Ed, sv, Vd, &Ud);
    V3 u0{Ud(0, 0), Ud(1, 0), Ud(2, 0)}, u1{Ud(0, 1), Ud(1, 1), Ud(2, 1)};
    u0 = normalized(u0);
    u1 = normalized(u1 - u0 * dot(u0, u1));
    V3 u2 = cross(u0, u1);   // the left null direction; det(U) = +1 by construction
    M3 U, VT;
    for (int r = 0; r < 3; ++r) {
        U(r, 0) = u0[r];
        U(r, 1) = u1[r];
        U(r, 2) = u2[r];
        for (int c = 0; c < 3; ++c) VT(r, c) = Vd(c, r);
    }
    if (det(VT) < 0) VT = scaled(VT, -1.0);
    M3 W;
    W(0, 1) = 1; W(1, 0) = -1; W(2, 2) = 1;
    R1 = U * W * VT;
    R2 = U * transpose(W) * VT;
    T = u2;
}

// two-view DLT; P = [R | T] row-major 3x4
inline std::array<double, 4> triangulate_point(const P34 &P1, const P34 &P2, V3 z1, V3 z2) {
    return triangulate_point(std::vector<P34>{P1, P2}, std::vector<V3>{z1, z2});
}

// ---------------------------------------------------------------------------------- small dense solves
// Minimum-residual solution of A x = b by Householder QR with column pivoting (rank-revealing; the columns past
// the numerical rank get zero).  Stands in for Eigen's fullPivHouseholderQr().solve(b).
inline std::vector<double> lstsq_qr(Dense A, std::vector<double> b) {
    const int m = A.r, n = A.c, k = std::min(m, n);
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::vector<double> diag(k, 0.0);
    int rank = 0;
    double biggest = 0.0;
    for (int j = 0; j < k; ++j) {
        int best = j;
        double best_n2 = -1.0;
        for (int c = j; c < n; ++c) {
            double n2 = 0;
            for (int r = j; r < m; ++r) n2 += A(r, c) * A(r, c);
            if (n2 > best_n2) {
                best_n2 = n2;
                best = c;
            }
        }
        if (best != j) {
            for (int r = 0; r < m; ++r) std::swap(A(r, j), A(r, best));
            std::swap(perm[j], perm[best]);
        }
        double alpha = std::sqrt(best_n2);
        if (j == 0) biggest = alpha;
        if (alpha <= biggest * std::numeric_limits<double>::epsilon() * std::max(m, n)) break;
        if (A(j, j) > 0) alpha = -alpha;
        // v = x - alpha e1, H = I - 2 v v^T / (v^T v)
        std::vector<double> v(m - j);
        for (int r = j; r < m; ++r) v[r - j] = A(r, j);
        v[0] -= alpha;
        double vv = 0;
        for (double t : v) vv += t * t;
        if (vv > 0) {
            for (int c = j; c < n; ++c) {
                double s = 0;
                for (int r = j; r < m; ++r) s += v[r - j] * A(r, c);
                s = 2.0 * s / vv;
                for (int r = j; r < m; ++r) A(r, c) -= s * v[r - j];
            }
            double s = 0;
            for (int r = j; r < m; ++r) s += v[r - j] * b[r];
            s = 2.0 * s / vv;
            for (int r = j; r < m; ++r) b[r] -= s * v[r - j];
        }
        diag[j] = A(j, j);
        rank = j + 1;
    }
    std::vector<double> y(n, 0.0), x(n, 0.0);
    for (int i = rank - 1; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < rank; ++c) s -= A(i, c) * y[c];
        y[i] = s / A(i, i);
    }
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    return x;
}

// x = V S^+ U^T b for a 3x3 system (JacobiSVD<Matrix3d>::solve): singular values below eps * 3 * s_max are dropped
inline V3 svd_solve3(const M3 &A, V3 b) {
    Dense Ad(3, 3), V, U;
    for (int i = 0; i < 9; ++i) Ad.a[i] = A.m[i];
    std::vector<double> s;
    jacobi_svd(Ad, s, V, &U);
    V3 x{0, 0, 0};
    for (int j = 0; j < 3; ++j) {
        if (!(s[j] > s[0] * 3 * std::numeric_limits<double>::epsilon())) continue;
        double c = (U(0, j) * b.x + U(1, j) * b.y + U(2, j) * b.z) / s[j];
        x = x + V3{V(0, j), V(1, j), V(2, j)} * c;
    }
    return x;
}

}   // namespace xrh
