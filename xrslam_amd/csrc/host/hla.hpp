// hla.hpp -- small host-side linear algebra for the order-defining pipeline logic
// (two-view RANSAC gates, triangulation, pose bookkeeping).  The heavy arithmetic of the hot
// path runs on the GPU; what is here operates on a handful of 3-vectors per frame, exactly the
// work the reference keeps on the host with Eigen (which is not available in this image).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <vector>

namespace xrh {

struct V2 {
    double x = 0, y = 0;
};
struct V3 {
    double x = 0, y = 0, z = 0;
    double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return a * s; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) {
    double n2 = dot(a, a);
    return n2 > 0 ? a / std::sqrt(n2) : a;
}
inline V3 stable_normalized(V3 v) {
    double w = std::max(std::fabs(v.x), std::max(std::fabs(v.y), std::fabs(v.z)));
    V3 s = v / w;
    double z = dot(s, s);
    if (z > 0) return v / (std::sqrt(z) * w);
    return v;
}

struct M3 {
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double &operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
    static M3 identity() {
        M3 r;
        r.m[0] = r.m[4] = r.m[8] = 1;
        return r;
    }
};
inline M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
inline V3 operator*(const M3 &a, V3 v) {
    return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
            a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z};
}
inline M3 transpose(const M3 &a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
    return r;
}
inline double det(const M3 &a) {
    return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
           a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
}

struct Quat {   // x,y,z,w like Eigen coeffs()
    double x = 0, y = 0, z = 0, w = 1;
    Quat conjugate() const { return {-x, -y, -z, w}; }
    Quat normalized() const {
        double n = std::sqrt(x * x + y * y + z * z + w * w);
        return {x / n, y / n, z / n, w / n};
    }
};
inline Quat operator*(Quat a, Quat b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline V3 operator*(Quat q, V3 v) {   // Eigen _transformVector
    V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + uv * q.w + cross(u, uv);
}
inline M3 to_matrix(Quat q) {
    M3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
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
inline Quat expmap(V3 w) {   // geometry/lie_algebra.h:13-18
    double angle = norm(w);
    V3 axis = stable_normalized(w);
    double ha = 0.5 * angle, s = std::sin(ha);
    return {s * axis.x, s * axis.y, s * axis.z, std::cos(ha)};
}

// ------------------------------------------------------------------------------------------
// dense helpers on std::vector<double>, row-major
struct Dense {
    int r = 0, c = 0;
    std::vector<double> a;
    Dense() {}
    Dense(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

// One-sided Jacobi SVD  A (m x n) = U diag(s) V^T.  Returns V (n x n) and singular values sorted
// descending; columns of V belonging to (numerically) zero singular values span the null space.
// Plays the role of Eigen::JacobiSVD(ComputeFullV) in stereo.h:84-94, essential.cpp:105-117, wahba.h:17-18.
//
// jacobi_sweeps is the rotation loop on caller-provided row-major storage (A: m x n, V: n x n = identity on entry); MT / NT are the
// dimensions as compile-time constants (0: taken from m / n at run time).  The shapes the per-frame path meets -- the 5 x 9 epipolar
// system of the 5-point solver, the 3 x 3 covariance of the 2-point rotation -- get instances whose loops the compiler unrolls; every
// instance executes the same floating-point operations in the same order, so the result does not depend on the instance.
template <int MT, int NT> inline void jacobi_sweeps(double *A, double *V, int m_rt, int n_rt) {
    const int m = MT ? MT : m_rt, n = NT ? NT : n_rt;
    double fro2 = 0.0;
    for (int i = 0; i < m * n; ++i) fro2 += A[i] * A[i];
    // a column whose norm has dropped below 1e-15 |A|_F is a converged null-space direction: rotating it further
    // only chases rounding noise (rank-deficient inputs -- the 5x9 epipolar system, a 2-point covariance -- would
    // otherwise burn every sweep on such columns)
    const double null2 = 1e-30 * fro2;
    // squared column norms are recomputed only for columns a rotation has touched since they were last formed (the same sum of the
    // same entries gives the same value: a cached norm IS the recomputed one)
    constexpr int NC = NT ? NT : 1;
    double st_norm[NC];
    bool st_ok[NC];
    std::vector<double> heap_norm;
    std::vector<char> heap_ok;
    double *norm2 = st_norm;
    bool *have = st_ok;
    if (!NT) {
        heap_norm.resize(n);
        heap_ok.resize(n);
        norm2 = heap_norm.data();
        have = reinterpret_cast<bool *>(heap_ok.data());
    }
    for (int j = 0; j < n; ++j) have[j] = false;
    auto col_norm2 = [&](int j) {
        if (!have[j]) {
            double t = 0;
            for (int i = 0; i < m; ++i) t += A[i * n + j] * A[i * n + j];
            norm2[j] = t;
            have[j] = true;
        }
        return norm2[j];
    };
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double al = col_norm2(p), be = col_norm2(q);
                double ga = 0;
                for (int i = 0; i < m; ++i) ga += A[i * n + p] * A[i * n + q];
                if (ga == 0.0 || std::fabs(ga) <= 1e-16 * std::sqrt(al * be) || std::min(al, be) <= null2) continue;
                rotated = true;
                double zeta = (be - al) / (2.0 * ga);
                double t = 1.0 / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                if (zeta < 0) t = -t;
                double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < m; ++i) {
                    double x = A[i * n + p], y = A[i * n + q];
                    A[i * n + p] = cs * x - sn * y;
                    A[i * n + q] = sn * x + cs * y;
                }
                have[p] = have[q] = false;
                for (int i = 0; i < n; ++i) {
                    double x = V[i * n + p], y = V[i * n + q];
                    V[i * n + p] = cs * x - sn * y;
                    V[i * n + q] = sn * x + cs * y;
                }
            }
        if (!rotated) break;
    }
}
inline void jacobi_svd(const Dense &A_in, std::vector<double> &s, Dense &V, Dense *U = nullptr) {
    const int m = A_in.r, n = A_in.c;
    Dense A = A_in;
    V = Dense(n, n);
    for (int i = 0; i < n; ++i) V(i, i) = 1.0;
    if (m == 5 && n == 9) jacobi_sweeps<5, 9>(A.a.data(), V.a.data(), m, n);
    else if (m == 3 && n == 3) jacobi_sweeps<3, 3>(A.a.data(), V.a.data(), m, n);
    else jacobi_sweeps<0, 0>(A.a.data(), V.a.data(), m, n);
    std::vector<double> nrm(n);
    std::vector<int> idx(n);
    for (int j = 0; j < n; ++j) {
        double t = 0;
        for (int i = 0; i < m; ++i) t += A(i, j) * A(i, j);
        nrm[j] = std::sqrt(t);
        idx[j] = j;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
    Dense Vs(n, n);
    s.resize(n);
    if (U) *U = Dense(m, n);
    for (int j = 0; j < n; ++j) {
        s[j] = nrm[idx[j]];
        for (int i = 0; i < n; ++i) Vs(i, j) = V(i, idx[j]);
        if (U)
            for (int i = 0; i < m; ++i) (*U)(i, j) = s[j] > 0 ? A(i, idx[j]) / s[j] : 0.0;
    }
    V = Vs;
}

// Eigenvalues of a general real n x n matrix (Hessenberg reduction + shifted QR, EISPACK elmhes/hqr scheme),
// and for every real eigenvalue an eigenvector by inverse iteration.  Plays the role of
// Eigen::EigenSolver in essential.cpp:202-218.
// real_eigen_core<NT>: the routine on row-major storage, NT = n as a compile-time constant (0: run time).  The 10 x 10 action matrix
// of the 5-point solver gets an instance with stack storage and unrolled loops; every instance performs the same operations in the
// same order.
template <int NT>
inline void real_eigen_core(const double *Mbuf, int n_rt, std::vector<double> &wr, std::vector<double> &wi,
                            std::vector<std::vector<double>> &vecs) {
    const int n = NT ? NT : n_rt;
    double st_a[NT ? NT * NT : 1], st_b[NT ? NT * NT : 1], st_v[NT ? NT : 1];
    int st_piv[NT ? NT : 1];
    std::vector<double> heap_d;
    std::vector<int> heap_i;
    double *abuf = st_a, *bbuf = st_b, *v = st_v;
    int *piv = st_piv;
    if (!NT) {
        heap_d.resize((size_t)2 * n * n + n);
        heap_i.resize(n);
        abuf = heap_d.data();
        bbuf = abuf + (size_t)n * n;
        v = bbuf + (size_t)n * n;
        piv = heap_i.data();
    }
    for (int k = 0; k < n * n; ++k) abuf[k] = Mbuf[k];
    auto a = [=](int i, int j) -> double & { return abuf[i * n + j]; };
    auto B = [=](int i, int j) -> double & { return bbuf[i * n + j]; };
    auto M = [=](int i, int j) -> double { return Mbuf[i * n + j]; };
    for (int k = 0; k < n * n; ++k)
        if (!std::isfinite(Mbuf[k])) {   // Eigen::EigenSolver yields NaN eigenvalues here; no real eigenpair is reported
            wr.assign(n, std::numeric_limits<double>::quiet_NaN());
            wi.assign(n, std::numeric_limits<double>::quiet_NaN());
            vecs.assign(n, std::vector<double>());
            return;
        }
    // balance-free Hessenberg reduction by stabilised elementary similarity transforms
    for (int m = 1; m < n - 1; ++m) {
        double x = 0.0;
        int i = m;
        for (int j = m; j < n; ++j)
            if (std::fabs(a(j, m - 1)) > std::fabs(x)) {
                x = a(j, m - 1);
                i = j;
            }
        if (i != m) {
            for (int j = m - 1; j < n; ++j) std::swap(a(i, j), a(m, j));
            for (int j = 0; j < n; ++j) std::swap(a(j, i), a(j, m));
        }
        if (x != 0.0)
            for (i = m + 1; i < n; ++i) {
                double y = a(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    a(i, m - 1) = y;
                    for (int j = m; j < n; ++j) a(i, j) -= y * a(m, j);
                    for (int j = 0; j < n; ++j) a(j, m) += y * a(j, i);
                }
            }
    }
    for (int i = 2; i < n; ++i)
        for (int j = 0; j < i - 1; ++j) a(i, j) = 0.0;
    wr.assign(n, 0.0);
    wi.assign(n, 0.0);
    int nn = n - 1, its = 0;
    double t = 0.0, p = 0, q = 0, r = 0, s = 0, w = 0, x = 0, y = 0, z = 0;
    double anorm = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = std::max(i - 1, 0); j < n; ++j) anorm += std::fabs(a(i, j));
    while (nn >= 0) {
        its = 0;
        int l;
        do {
            for (l = nn; l >= 1; --l) {
                s = std::fabs(a(l - 1, l - 1)) + std::fabs(a(l, l));
                if (s == 0.0) s = anorm;
                if (std::fabs(a(l, l - 1)) + s == s) {
                    a(l, l - 1) = 0.0;
                    break;
                }
            }
            x = a(nn, nn);
            if (l == nn) {
                wr[nn] = x + t;
                wi[nn--] = 0.0;
            } else {
                y = a(nn - 1, nn - 1);
                w = a(nn, nn - 1) * a(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = std::sqrt(std::fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + (p >= 0 ? std::fabs(z) : -std::fabs(z));
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn - 1] = -(wi[nn] = z);
                    }
                    nn -= 2;
                } else {
                    if (its == 60) return;   // no convergence: leave remaining eigenvalues at zero
                    if (its == 10 || its == 20) {
                        t += x;
                        for (int i = 0; i <= nn; ++i) a(i, i) -= x;
                        s = std::fabs(a(nn, nn - 1)) + std::fabs(a(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = a(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / a(m + 1, m) + a(m, m + 1);
                        q = a(m + 1, m + 1) - z - r - s;
                        r = a(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s;
                        q /= s;
                        r /= s;
                        if (m == l) break;
                        double u = std::fabs(a(m, m - 1)) * (std::fabs(q) + std::fabs(r));
                        double v = std::fabs(p) * (std::fabs(a(m - 1, m - 1)) + std::fabs(z) + std::fabs(a(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        a(i, i - 2) = 0.0;
                        if (i != m + 2) a(i, i - 3) = 0.0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = a(k, k - 1);
                            q = a(k + 1, k - 1);
                            r = 0.0;
                            if (k != nn - 1) r = a(k + 2, k - 1);
                            if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) {
                                p /= x;
                                q /= x;
                                r /= x;
                            }
                        }
                        double sg = std::sqrt(p * p + q * q + r * r);
                        s = p >= 0 ? sg : -sg;
                        if (s != 0.0) {
                            if (k == m) {
                                if (l != m) a(k, k - 1) = -a(k, k - 1);
                            } else {
                                a(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s;
                            y = q / s;
                            z = r / s;
                            q /= p;
                            r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = a(k, j) + q * a(k + 1, j);
                                if (k != nn - 1) {
                                    p += r * a(k + 2, j);
                                    a(k + 2, j) -= p * z;
                                }
                                a(k + 1, j) -= p * y;
                                a(k, j) -= p * x;
                            }
                            int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * a(i, k) + y * a(i, k + 1);
                                if (k != nn - 1) {
                                    p += z * a(i, k + 2);
                                    a(i, k + 2) -= p * r;
                                }
                                a(i, k + 1) -= p * q;
                                a(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    // eigenvectors of the real eigenvalues by inverse iteration on the ORIGINAL matrix
    vecs.assign(n, std::vector<double>());
    for (int e = 0; e < n; ++e) {
        if (std::fabs(wi[e]) >= 1.0e-10) continue;
        const double lam = wr[e];
        double scale = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                B(i, j) = M(i, j) - (i == j ? lam : 0.0);
                scale = std::max(scale, std::fabs(M(i, j)));
            }
        const double eps = std::max(scale, 1.0) * 1e-13;
        for (int i = 0; i < n; ++i) B(i, i) += eps * (1 + i % 3);   // keep the shifted matrix invertible
        // LU with partial pivoting
        for (int k = 0; k < n; ++k) {
            int pk = k;
            for (int i = k + 1; i < n; ++i)
                if (std::fabs(B(i, k)) > std::fabs(B(pk, k))) pk = i;
            piv[k] = pk;
            if (pk != k)
                for (int j = 0; j < n; ++j) std::swap(B(k, j), B(pk, j));
            if (B(k, k) == 0.0) B(k, k) = eps;
            for (int i = k + 1; i < n; ++i) {
                B(i, k) /= B(k, k);
                for (int j = k + 1; j < n; ++j) B(i, j) -= B(i, k) * B(k, j);
            }
        }
        for (int k = 0; k < n; ++k) v[k] = 1.0;
        for (int it = 0; it < 4; ++it) {
            for (int k = 0; k < n; ++k) {
                if (piv[k] != k) std::swap(v[k], v[piv[k]]);
                for (int i = k + 1; i < n; ++i) v[i] -= B(i, k) * v[k];
            }
            for (int i = n - 1; i >= 0; --i) {
                for (int j = i + 1; j < n; ++j) v[i] -= B(i, j) * v[j];
                v[i] /= B(i, i);
            }
            double nrm = 0;
            for (int k = 0; k < n; ++k) nrm += v[k] * v[k];
            nrm = std::sqrt(nrm);
            if (!(nrm > 0) || !std::isfinite(nrm)) break;
            for (int k = 0; k < n; ++k) v[k] /= nrm;
        }
        vecs[e].assign(v, v + n);
    }
}
inline void real_eigen(const Dense &M, std::vector<double> &wr, std::vector<double> &wi,
                       std::vector<std::vector<double>> &vecs) {
    if (M.r == 10) real_eigen_core<10>(M.a.data(), M.r, wr, wi, vecs);
    else real_eigen_core<0>(M.a.data(), M.r, wr, wi, vecs);
}

}   // namespace xrh
