// la.hpp -- tiny dense linear algebra for the CPU oracle (test infrastructure only).
// Restates the Eigen 3.3.7 semantics the reference relies on (quaternion
// product/rotation, AngleAxis conversions, stableNormalized, LLT, inverse).
// Eigen itself is not available in this image.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

template <int R, int C> struct Mat {
    double a[R * C];
    Mat() { std::memset(a, 0, sizeof(a)); }
    double &operator()(int r, int c) { return a[r * C + c]; }
    double operator()(int r, int c) const { return a[r * C + c]; }
    double &operator[](int i) { return a[i]; }
    double operator[](int i) const { return a[i]; }
    static Mat identity() {
        Mat m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
        return m;
    }
    Mat<C, R> t() const {
        Mat<C, R> m;
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < C; ++c) m(c, r) = (*this)(r, c);
        return m;
    }
    Mat operator+(const Mat &o) const {
        Mat m;
        for (int i = 0; i < R * C; ++i) m.a[i] = a[i] + o.a[i];
        return m;
    }
    Mat operator-(const Mat &o) const {
        Mat m;
        for (int i = 0; i < R * C; ++i) m.a[i] = a[i] - o.a[i];
        return m;
    }
    Mat operator-() const {
        Mat m;
        for (int i = 0; i < R * C; ++i) m.a[i] = -a[i];
        return m;
    }
    Mat operator*(double s) const {
        Mat m;
        for (int i = 0; i < R * C; ++i) m.a[i] = a[i] * s;
        return m;
    }
    Mat operator/(double s) const {
        Mat m;
        for (int i = 0; i < R * C; ++i) m.a[i] = a[i] / s;
        return m;
    }
    Mat &operator+=(const Mat &o) {
        for (int i = 0; i < R * C; ++i) a[i] += o.a[i];
        return *this;
    }
    Mat &operator-=(const Mat &o) {
        for (int i = 0; i < R * C; ++i) a[i] -= o.a[i];
        return *this;
    }
    double squaredNorm() const {
        double s = 0;
        for (int i = 0; i < R * C; ++i) s += a[i] * a[i];
        return s;
    }
    double norm() const { return std::sqrt(squaredNorm()); }
    double dot(const Mat &o) const {
        double s = 0;
        for (int i = 0; i < R * C; ++i) s += a[i] * o.a[i];
        return s;
    }
    template <int BR, int BC> Mat<BR, BC> block(int r0, int c0) const {
        Mat<BR, BC> m;
        for (int r = 0; r < BR; ++r)
            for (int c = 0; c < BC; ++c) m(r, c) = (*this)(r0 + r, c0 + c);
        return m;
    }
    template <int BR, int BC> void set_block(int r0, int c0, const Mat<BR, BC> &b) {
        for (int r = 0; r < BR; ++r)
            for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) = b(r, c);
    }
    template <int BR, int BC> void add_block(int r0, int c0, const Mat<BR, BC> &b) {
        for (int r = 0; r < BR; ++r)
            for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) += b(r, c);
    }
};

template <int R, int K, int C> Mat<R, C> operator*(const Mat<R, K> &x, const Mat<K, C> &y) {
    Mat<R, C> m;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += x(r, k) * y(k, c);
            m(r, c) = s;
        }
    return m;
}
template <int R, int C> Mat<R, C> operator*(double s, const Mat<R, C> &m) { return m * s; }

using Vec3 = Mat<3, 1>;
using Vec2 = Mat<2, 1>;
using Mat3 = Mat<3, 3>;

inline Vec3 vec3(double x, double y, double z) {
    Vec3 v;
    v[0] = x;
    v[1] = y;
    v[2] = z;
    return v;
}
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline Mat3 hat(const Vec3 &w) {   // geometry/lie_algebra.h:8-11
    Mat3 m;
    m(0, 1) = -w[2];
    m(0, 2) = w[1];
    m(1, 0) = w[2];
    m(1, 2) = -w[0];
    m(2, 0) = -w[1];
    m(2, 1) = w[0];
    return m;
}
inline Vec3 normalized(const Vec3 &v) {
    double n2 = v.squaredNorm();
    if (n2 > 0) return v / std::sqrt(n2);
    return v;
}
inline Vec3 stable_normalized(const Vec3 &v) {   // Eigen MatrixBase::stableNormalized
    double w = std::max(std::fabs(v[0]), std::max(std::fabs(v[1]), std::fabs(v[2])));
    Vec3 s = v / w;
    double z = s.squaredNorm();
    if (z > 0) return v / (std::sqrt(z) * w);
    return v;
}

struct Quat {   // storage order x,y,z,w like Eigen::Quaternion::coeffs()
    double x = 0, y = 0, z = 0, w = 1;
    Quat() {}
    Quat(double w_, double x_, double y_, double z_) : x(x_), y(y_), z(z_), w(w_) {}
    static Quat from_xyzw(const double *p) { return Quat(p[3], p[0], p[1], p[2]); }
    void to_xyzw(double *p) const {
        p[0] = x;
        p[1] = y;
        p[2] = z;
        p[3] = w;
    }
    Vec3 vec() const { return vec3(x, y, z); }
    Quat conjugate() const { return Quat(w, -x, -y, -z); }
    double norm() const { return std::sqrt(x * x + y * y + z * z + w * w); }
    Quat normalized() const {
        double n = norm();
        return Quat(w / n, x / n, y / n, z / n);
    }
    Quat operator*(const Quat &b) const {   // Eigen quat_product
        return Quat(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                    w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x);
    }
    Vec3 operator*(const Vec3 &v) const {   // Eigen QuaternionBase::_transformVector
        Vec3 u = vec();
        Vec3 uv = cross(u, v);
        uv = uv + uv;
        return v + uv * w + cross(u, uv);
    }
    Mat3 matrix() const {   // Eigen toRotationMatrix
        Mat3 r;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        r(0, 0) = 1 - (tyy + tzz);
        r(0, 1) = txy - twz;
        r(0, 2) = txz + twy;
        r(1, 0) = txy + twz;
        r(1, 1) = 1 - (txx + tzz);
        r(1, 2) = tyz - twx;
        r(2, 0) = txz - twy;
        r(2, 1) = tyz + twx;
        r(2, 2) = 1 - (txx + tyy);
        return r;
    }
};

inline Quat expmap(const Vec3 &w) {   // geometry/lie_algebra.h:13-18 (AngleAxis -> Quaternion)
    double angle = w.norm();
    Vec3 axis = stable_normalized(w);
    double ha = 0.5 * angle;
    double s = std::sin(ha);
    return Quat(std::cos(ha), s * axis[0], s * axis[1], s * axis[2]);
}
inline Vec3 logmap(const Quat &q) {   // geometry/lie_algebra.h:20-23 (Quaternion -> AngleAxis)
    Vec3 v = q.vec();
    double n = v.norm();
    if (n < std::numeric_limits<double>::epsilon()) {
        double m = std::max(std::fabs(v[0]), std::max(std::fabs(v[1]), std::fabs(v[2])));   // stableNorm
        n = (m > 0) ? m * (v / m).norm() : 0.0;
    }
    if (n != 0.0) {
        double angle = 2.0 * std::atan2(n, std::fabs(q.w));
        if (q.w < 0) n = -n;
        return (v / n) * angle;
    }
    return vec3(1, 0, 0) * 0.0;
}

inline Mat3 right_jacobian(const Vec3 &w) {   // geometry/lie_algebra.cpp:5-45
    static const double root2_eps = std::sqrt(std::numeric_limits<double>::epsilon());
    static const double root4_eps = std::sqrt(root2_eps);
    static const double qdrt720 = std::sqrt(std::sqrt(720.0));
    static const double qdrt5040 = std::sqrt(std::sqrt(5040.0));
    static const double sqrt24 = std::sqrt(24.0);
    static const double sqrt120 = std::sqrt(120.0);
    double angle = w.norm();
    double cangle = std::cos(angle), sangle = std::sin(angle);
    double angle2 = angle * angle;
    double cos_term;
    if (angle > root4_eps * qdrt720) {
        cos_term = (1 - cangle) / angle2;
    } else {
        cos_term = 0.5;
        if (angle > root2_eps * sqrt24) cos_term -= angle2 / 24.0;
    }
    double sin_term;
    if (angle > root4_eps * qdrt5040) {
        sin_term = (angle - sangle) / (angle * angle2);
    } else {
        sin_term = 1.0 / 6.0;
        if (angle > root2_eps * sqrt120) sin_term -= angle2 / 120.0;
    }
    Mat3 hw = hat(w);
    return Mat3::identity() - hw * cos_term + (hw * hw) * sin_term;
}

inline Mat3 inverse3(const Mat3 &m) {   // Eigen fixed 3x3 inverse: cofactors / determinant
    Mat3 c;
    c(0, 0) = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1);
    c(0, 1) = m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2);
    c(0, 2) = m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1);
    c(1, 0) = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2);
    c(1, 1) = m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0);
    c(1, 2) = m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2);
    c(2, 0) = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
    c(2, 1) = m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1);
    c(2, 2) = m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0);
    double det = m(0, 0) * c(0, 0) + m(0, 1) * c(1, 0) + m(0, 2) * c(2, 0);
    return c * (1.0 / det);
}

// s2_tangential_basis (geometry/lie_algebra.cpp:47-56): columns b1,b2
inline void s2_tangential_basis(const Vec3 &x, Vec3 &b1, Vec3 &b2) {
    int d = 0;
    for (int i = 1; i < 3; ++i)
        if (std::fabs(x[i]) > std::fabs(x[d])) d = i;
    Vec3 e;
    e[(d + 1) % 3] = 1.0;
    b1 = normalized(cross(x, e));
    b2 = normalized(cross(x, b1));
}

// ---------------------------------------------------------------- dynamic dense
struct DMat {   // row-major
    int r = 0, c = 0;
    std::vector<double> a;
    DMat() {}
    DMat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

// in-place inverse by LU with partial pivoting (Eigen PartialPivLU::inverse semantics); returns false if singular
inline bool invert(DMat &m) {
    int n = m.r;
    DMat inv(n, n);
    for (int i = 0; i < n; ++i) inv(i, i) = 1.0;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(m(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(m(i, k)) > best) {
                best = std::fabs(m(i, k));
                p = i;
            }
        if (best == 0.0) return false;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                std::swap(m(k, j), m(p, j));
                std::swap(inv(k, j), inv(p, j));
            }
        double d = m(k, k);
        for (int i = k + 1; i < n; ++i) {
            double f = m(i, k) / d;
            if (f == 0.0) continue;
            for (int j = k; j < n; ++j) m(i, j) -= f * m(k, j);
            for (int j = 0; j < n; ++j) inv(i, j) -= f * inv(k, j);
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double d = m(k, k);
        for (int j = 0; j < n; ++j) inv(k, j) /= d;
        for (int i = 0; i < k; ++i) {
            double f = m(i, k);
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) inv(i, j) -= f * inv(k, j);
        }
    }
    m = inv;
    return true;
}

// Cholesky M = L L^T (lower), returns false if not positive definite
inline bool cholesky_lower(const DMat &m, DMat &L) {
    int n = m.r;
    L = DMat(n, n);
    for (int j = 0; j < n; ++j) {
        double d = m(j, j);
        for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        L(j, j) = d;
        for (int i = j + 1; i < n; ++i) {
            double s = m(i, j);
            for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
            L(i, j) = s / d;
        }
    }
    return true;
}

// symmetric eigen-decomposition by cyclic Jacobi: A = V diag(w) V^T, w ascending
inline void sym_eigen(const DMat &A_in, std::vector<double> &w, DMat &V) {
    int n = A_in.r;
    DMat A = A_in;
    V = DMat(n, n);
    for (int i = 0; i < n; ++i) V(i, i) = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) {
            diag += A(i, i) * A(i, i);
            for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
        }
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A(p, q);
                if (apq == 0.0) continue;
                double app = A(p, p), aqq = A(q, q);
                if (std::fabs(apq) < 1e-300) continue;
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                if (!std::isfinite(theta)) t = 0.0;
                double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    double akp = A(k, p), akq = A(k, q);
                    A(k, p) = cs * akp - sn * akq;
                    A(k, q) = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A(p, k), aqk = A(q, k);
                    A(p, k) = cs * apk - sn * aqk;
                    A(q, k) = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V(k, p), vkq = V(k, q);
                    V(k, p) = cs * vkp - sn * vkq;
                    V(k, q) = sn * vkp + cs * vkq;
                }
            }
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return A(a, a) < A(b, b); });
    w.resize(n);
    DMat Vs(n, n);
    for (int j = 0; j < n; ++j) {
        w[j] = A(idx[j], idx[j]);
        for (int i = 0; i < n; ++i) Vs(i, j) = V(i, idx[j]);
    }
    V = Vs;
}

}   // namespace orc
