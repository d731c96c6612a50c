// geometry.hpp -- two-view gates and triangulation used by Frame::track_keypoints / Track::triangulate.
//
// Host-side mirror of (file:line under /root/reference/xrslam/src/xrslam):
//   apply_k / remove_k / triangulate_point       geometry/stereo.h:8-21, 84-94
//   find_essential_matrix / find_rotation_matrix geometry/stereo.cpp:38-91
//   solve_essential_5pt                          geometry/essential.cpp:105-297 (Nister/Stewenius action matrix)
//   essential_geometric_error                    geometry/essential.h:15-20
//   solve_rotation_2pt                           geometry/wahba.h:9-27
//   Ransac<>::solve                              utility/ransac.h:27-83
//   LotBox                                       utility/random.h:80-126 (std::default_random_engine)
// These run once per frame on <= a few hundred points and decide inlier masks / flags that the
// rest of the pipeline branches on, so they stay sequential on the host like in the reference.
#pragma once
#include <array>
#include <cmath>
#include <map>
#include <numeric>
#include <random>
#include <vector>

#include "hla.hpp"

namespace xrh {

struct Intrinsics {
    double fx = 1, fy = 1, cx = 0, cy = 0;
};

inline V2 apply_k(V3 p, const Intrinsics &K) { return {p.x / p.z * K.fx + K.cx, p.y / p.z * K.fy + K.cy}; }
inline V3 remove_k(V2 p, const Intrinsics &K) { return normalized(V3{(p.x - K.cx) / K.fx, (p.y - K.cy) / K.fy, 1.0}); }

// N-view DLT: rows  z.x P.row(2) - z.z P.row(0),  z.y P.row(2) - z.z P.row(1); null vector by SVD
struct P34 {
    double m[12];
};
inline std::array<double, 4> triangulate_point(const std::vector<P34> &Ps, const std::vector<V3> &zs) {
    Dense A((int)zs.size() * 2, 4);
    for (size_t i = 0; i < zs.size(); ++i)
        for (int c = 0; c < 4; ++c) {
            A((int)i * 2, c) = zs[i].x * Ps[i].m[8 + c] - zs[i].z * Ps[i].m[c];
            A((int)i * 2 + 1, c) = zs[i].y * Ps[i].m[8 + c] - zs[i].z * Ps[i].m[4 + c];
        }
    std::vector<double> s;
    Dense V;
    jacobi_svd(A, s, V);
    return {V(0, 3), V(1, 3), V(2, 3), V(3, 3)};
}

// ---------------------------------------------------------------------------------- LotBox
class LotBox {
  public:
    explicit LotBox(size_t size) : cap_(0), lots_(size) { std::iota(lots_.begin(), lots_.end(), 0); }
    void seed(unsigned v) { engine_.seed(v); }
    void refill_all() { cap_ = 0; }
    size_t remaining() const { return lots_.size() - cap_; }
    size_t draw_without_replacement() {
        if (remaining() > 1) {
            std::uniform_int_distribution<size_t>::param_type pr(cap_, lots_.size() - 1);
            std::swap(lots_[cap_], lots_[dist_(engine_, pr)]);
            return lots_[cap_++];
        } else if (remaining() == 1) {
            cap_++;
            return lots_.back();
        }
        return size_t(-1);
    }

  private:
    size_t cap_;
    std::vector<size_t> lots_;
    std::default_random_engine engine_;
    std::uniform_int_distribution<size_t> dist_;
};

// ------------------------------------------------------------------- 5-point essential solver
namespace fivept {

// Polynomials in (x,y,z) of a known degree bound D <= 3, stored compactly: the coefficient of x^ex y^ey z^ez sits at the rank of its
// dense index ex*16 + ey*4 + ez among the monomials of degree <= D (SUP<D>: ascending dense index).  The entries of E are linear,
// E E^T quadratic, the ten constraints cubic.
// Arithmetic contract: a product accumulates its terms in the order of a dense double loop over the two factors' coefficient arrays
// (i ascending, then j ascending, zero coefficients skipped), a sum is a + sb * b coefficient by coefficient -- the operations and
// the order the solver has had since round 1 (dense 64-slot arrays then: the polynomial algebra was 17 of the solver's 40 us; the
// results are bit-identical, tests/test_host_geometry.py pins them).
inline constexpr int pidx(int ex, int ey, int ez) { return ex * 16 + ey * 4 + ez; }
template <int D> struct Sup;
template <> struct Sup<1> {
    static constexpr int N = 4;
    static constexpr unsigned char idx[4] = {0, 1, 4, 16};
};
template <> struct Sup<2> {
    static constexpr int N = 10;
    static constexpr unsigned char idx[10] = {0, 1, 2, 4, 5, 8, 16, 17, 20, 32};
};
template <> struct Sup<3> {
    static constexpr int N = 20;
    static constexpr unsigned char idx[20] = {0, 1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 17, 18, 20, 21, 24, 32, 33, 36, 48};
};
template <int D> inline constexpr int rank_of(int dense) {   // position of a dense index in Sup<D>::idx
    for (int k = 0; k < Sup<D>::N; ++k)
        if (Sup<D>::idx[k] == dense) return k;
    return -1;
}
template <int D> struct Poly {
    double c[Sup<D>::N];
    Poly() {
        for (double &v : c) v = 0.0;
    }
    double at(int ex, int ey, int ez) const { return c[rank_of<D>(pidx(ex, ey, ez))]; }
    double &at(int ex, int ey, int ez) { return c[rank_of<D>(pidx(ex, ey, ez))]; }
};
template <int D> inline Poly<D> padd(const Poly<D> &a, const Poly<D> &b, double sb = 1.0) {
    Poly<D> r;
    for (int k = 0; k < Sup<D>::N; ++k) r.c[k] = a.c[k] + sb * b.c[k];
    return r;
}
template <int D> inline Poly<D> pscale(const Poly<D> &a, double s) {
    Poly<D> r;
    for (int k = 0; k < Sup<D>::N; ++k) r.c[k] = a.c[k] * s;
    return r;
}
template <int DA, int DB> struct MulTable {   // rank of monomial(p) * monomial(q) in Sup<DA + DB> (dense indices add: no exponent exceeds 3)
    unsigned char t[Sup<DA>::N][Sup<DB>::N];
    constexpr MulTable() : t() {
        for (int p = 0; p < Sup<DA>::N; ++p)
            for (int q = 0; q < Sup<DB>::N; ++q) t[p][q] = (unsigned char)rank_of<DA + DB>(Sup<DA>::idx[p] + Sup<DB>::idx[q]);
    }
};
template <int DA, int DB> inline Poly<DA + DB> pmul(const Poly<DA> &a, const Poly<DB> &b) {
    static constexpr MulTable<DA, DB> T{};
    Poly<DA + DB> r;
    for (int p = 0; p < Sup<DA>::N; ++p) {
        const double av = a.c[p];
        if (av == 0.0) continue;
        for (int q = 0; q < Sup<DB>::N; ++q) {
            const double bv = b.c[q];
            if (bv == 0.0) continue;
            r.c[T.t[p][q]] += av * bv;
        }
    }
    return r;
}

// monomial order of the reference (GRevLex): xxx xxy xyy yyy xxz xyz yyz xzz yzz zzz xx xy yy xz yz zz x y z 1
static constexpr int MONO[20][3] = {{3, 0, 0}, {2, 1, 0}, {1, 2, 0}, {0, 3, 0}, {2, 0, 1}, {1, 1, 1}, {0, 2, 1},
                                {1, 0, 2}, {0, 1, 2}, {0, 0, 3}, {2, 0, 0}, {1, 1, 0}, {0, 2, 0}, {1, 0, 1},
                                {0, 1, 1}, {0, 0, 2}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
enum { XXX = 0, XXY, XYY, YYY, XXZ, XYZ, YYZ, XZZ, YZZ, ZZZ, XX, XY, YY, XZ, YZ, ZZ, X, Y, Z, I };
struct MonoRank {   // where the reference's m-th monomial sits in a Poly<3>
    unsigned char r[20];
    constexpr MonoRank() : r() {
        for (int m = 0; m < 20; ++m) r[m] = (unsigned char)rank_of<3>(pidx(MONO[m][0], MONO[m][1], MONO[m][2]));
    }
};

}   // namespace fivept

inline std::vector<M3> solve_essential_5pt(const std::array<V2, 5> &p1, const std::array<V2, 5> &p2) {
    using namespace fivept;
    for (int i = 0; i < 5; ++i)   // Eigen propagates NaNs to "no real eigenvalue"; short-circuit that case
        if (!std::isfinite(p1[i].x) || !std::isfinite(p1[i].y) || !std::isfinite(p2[i].x) || !std::isfinite(p2[i].y)) return {};
    // null space of the 5x9 epipolar constraint matrix
    Dense A(5, 9);
    for (int i = 0; i < 5; ++i) {
        const double h1[3] = {p1[i].x, p1[i].y, 1.0}, h2[3] = {p2[i].x, p2[i].y, 1.0};
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) A(i, j * 3 + k) = h1[j] * h2[k];
    }
    std::vector<double> sv;
    Dense V;
    jacobi_svd(A, sv, V);
    // basis = V.block<9,4>(0,5); E(x,y,z) = x Ex + y Ey + z Ez + Ew, to_matrix fills COLUMNS from consecutive triples
    double B[4][3][3];
    for (int b = 0; b < 4; ++b)
        for (int col = 0; col < 3; ++col)
            for (int row = 0; row < 3; ++row) B[b][row][col] = V(col * 3 + row, 5 + b);
    Poly<1> E[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            E[i][j].at(1, 0, 0) = B[0][i][j];
            E[i][j].at(0, 1, 0) = B[1][i][j];
            E[i][j].at(0, 0, 1) = B[2][i][j];
            E[i][j].at(0, 0, 0) = B[3][i][j];
        }
    Poly<2> EEt[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Poly<2> s;
            for (int k = 0; k < 3; ++k) s = padd(s, pmul(E[i][k], E[j][k]));
            EEt[i][j] = s;
        }
    const Poly<2> tr = padd(padd(EEt[0][0], EEt[1][1]), EEt[2][2]);
    const Poly<2> half_tr = pscale(tr, 0.5);
    Dense polys(10, 20);
    static constexpr MonoRank MR{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Poly<3> s;
            for (int k = 0; k < 3; ++k) s = padd(s, pmul(EEt[i][k], E[k][j]));
            s = padd(s, pmul(half_tr, E[i][j]), -1.0);
            for (int m = 0; m < 20; ++m) polys(i * 3 + j, m) = s.c[MR.r[m]];
        }
    {
        auto minor = [&](int r0, int c0, int r1, int c1, int r2, int c2, int r3, int c3) {   // E[r0][c0] E[r1][c1] - E[r2][c2] E[r3][c3]
            return padd(pmul(E[r0][c0], E[r1][c1]), pmul(E[r2][c2], E[r3][c3]), -1.0);
        };
        const Poly<3> d = padd(padd(pmul(E[0][0], minor(1, 1, 2, 2, 1, 2, 2, 1)), pmul(E[0][1], minor(1, 0, 2, 2, 1, 2, 2, 0)), -1.0),
                               pmul(E[0][2], minor(1, 0, 2, 1, 1, 1, 2, 0)));
        for (int m = 0; m < 20; ++m) polys(9, m) = d.c[MR.r[m]];
    }
    // Gauss-Jordan with the reference's row-permutation pivoting (essential.cpp:151-200)
    std::array<int, 10> perm;
    for (int i = 0; i < 10; ++i) perm[i] = i;
    for (int i = 0; i < 10; ++i) {
        for (int j = i + 1; j < 10; ++j)
            if (std::fabs(polys(perm[i], i)) < std::fabs(polys(perm[j], i))) std::swap(perm[i], perm[j]);
        if (polys(perm[i], i) == 0) continue;
        const double d = polys(perm[i], i);
        for (int c = 0; c < 20; ++c) polys(perm[i], c) /= d;
        for (int j = i + 1; j < 10; ++j) {
            const double f = polys(perm[j], i);
            for (int c = 0; c < 20; ++c) polys(perm[j], c) -= polys(perm[i], c) * f;
        }
    }
    for (int i = 9; i > 0; --i)
        for (int j = 0; j < i; ++j) {
            const double f = polys(perm[j], i);
            for (int c = 0; c < 20; ++c) polys(perm[j], c) -= polys(perm[i], c) * f;
        }
    Dense action(10, 10);
    const int rows[6] = {XXX, XXY, XYY, XXZ, XYZ, XZZ};
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 10; ++c) action(r, c) = -polys(perm[rows[r]], XX + c);
    action(6, XX - XX) = 1.0;
    action(7, XY - XX) = 1.0;
    action(8, XZ - XX) = 1.0;
    action(9, X - XX) = 1.0;
    std::vector<double> wr, wi;
    std::vector<std::vector<double>> vecs;
    real_eigen(action, wr, wi, vecs);
    std::vector<M3> out;
    for (int e = 0; e < 10; ++e) {
        if (vecs[e].empty()) continue;
        const std::vector<double> &h = vecs[e];
        const double w = h[I - XX];
        const double xs = h[X - XX] / w, ys = h[Y - XX] / w, zs = h[Z - XX] / w;
        M3 Em;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Em(i, j) = B[0][i][j] * xs + B[1][i][j] * ys + B[2][i][j] * zs + B[3][i][j];
        out.push_back(Em);
    }
    return out;
}

inline double essential_geometric_error(const M3 &E, V2 p1, V2 p2) {
    V3 Ep1 = E * V3{p1.x, p1.y, 1.0};
    double r = p2.x * Ep1.x + p2.y * Ep1.y + Ep1.z;
    return r * r / (Ep1.x * Ep1.x + Ep1.y * Ep1.y);
}

// h(p2) = R h(p1)
inline M3 solve_rotation_2pt(const std::array<V3, 2> &a, const std::array<V3, 2> &b) {
    Dense cov(3, 3);
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) cov(i, j) += a[k][i] * b[k][j];
    for (double &v : cov.a) v *= 0.5;
    std::vector<double> s;
    Dense V, U;
    jacobi_svd(cov, s, V, &U);
    // a rank-2 covariance leaves the third left singular vector undetermined: complete it to a right-handed frame
    V3 u0{U(0, 0), U(1, 0), U(2, 0)}, u1{U(0, 1), U(1, 1), U(2, 1)}, u2{U(0, 2), U(1, 2), U(2, 2)};
    if (!(s[2] > 1e-12 * std::max(s[0], 1e-300))) {
        u2 = normalized(cross(u0, u1));
        // keep V's third column as produced by the Jacobi rotations (orthonormal by construction)
    }
    M3 Um, Vm;
    for (int i = 0; i < 3; ++i) {
        Um(i, 0) = u0[i];
        Um(i, 1) = u1[i];
        Um(i, 2) = u2[i];
        for (int j = 0; j < 3; ++j) Vm(i, j) = V(i, j);
    }
    M3 D = M3::identity();
    D(2, 2) = det(Vm * transpose(Um)) >= 0.0 ? 1.0 : -1.0;
    return Vm * D * transpose(Um);
}

// ---------------------------------------------------------------------------------- RANSAC
template <size_t DoF, class Sample1, class Sample2, class Solver, class Evaluator>
M3 ransac_solve(const std::vector<Sample1> &d1, const std::vector<Sample2> &d2, double threshold, double confidence,
                size_t max_iteration, int seed, Solver solver, Evaluator make_eval, std::vector<char> &inlier_mask) {
    const size_t size = d1.size();
    LotBox lotbox(size);
    lotbox.seed((unsigned)seed);
    const double K = std::log(std::max(1 - confidence, 1.0e-5));
    size_t inlier_count = 0;
    M3 model;   // default-constructed like the reference's uninitialised ModelType (zeros here)
    if (size < DoF) {
        inlier_mask.assign(size, 0);
        return model;
    }
    size_t iter_max = max_iteration;
    std::vector<char> cur_mask;
    for (size_t iter = 0; iter < iter_max; ++iter) {
        std::array<Sample1, DoF> s1;
        std::array<Sample2, DoF> s2;
        lotbox.refill_all();
        for (size_t si = 0; si < DoF; ++si) {
            size_t idx = lotbox.draw_without_replacement();
            s1[si] = d1[idx];
            s2[si] = d2[idx];
        }
        std::vector<M3> models = solver(s1, s2);
        for (const M3 &cur : models) {
            // every sample is an inlier of the model held: no later model can count MORE (the update below is strict), and the
            // update that found it has already cut iter_max to zero -- nothing that follows changes the result
            if (inlier_count == size) break;
            size_t cur_count = 0;
            cur_mask.assign(size, 0);
            auto eval = make_eval(cur);
            for (size_t i = 0; i < size; ++i) {
                double err = eval(d1[i], d2[i]);
                if (err <= threshold) {
                    cur_count++;
                    cur_mask[i] = 1;
                }
            }
            if (cur_count > inlier_count) {
                model = cur;
                inlier_count = cur_count;
                inlier_mask.swap(cur_mask);
                double ratio = inlier_count / (double)size;
                double N = K / std::log(1 - std::pow(ratio, 5));
                if (N < (double)iter_max) iter_max = (size_t)std::ceil(N);
            }
        }
    }
    return model;
}

inline M3 find_essential_matrix(const std::vector<V2> &p1, const std::vector<V2> &p2, std::vector<char> &mask,
                                double threshold = 1.0, double confidence = 0.999, size_t max_iteration = 1000,
                                int seed = 0) {
    mask.clear();
    auto solver = [](const std::array<V2, 5> &a, const std::array<V2, 5> &b) { return solve_essential_5pt(a, b); };
    auto make_eval = [](const M3 &E) {
        M3 Et = transpose(E);
        return [E, Et](V2 a, V2 b) { return essential_geometric_error(E, a, b) + essential_geometric_error(Et, b, a); };
    };
    return ransac_solve<5>(p1, p2, 2.0 * 3.84 * threshold * threshold, confidence, max_iteration, seed, solver,
                           make_eval, mask);
}

inline M3 find_rotation_matrix(const std::vector<V3> &p1, const std::vector<V3> &p2, std::vector<char> &mask,
                               double threshold, double confidence = 0.999, size_t max_iteration = 1000,
                               int seed = 0) {
    mask.clear();
    auto solver = [](const std::array<V3, 2> &a, const std::array<V3, 2> &b) {
        return std::vector<M3>{solve_rotation_2pt(a, b)};
    };
    auto make_eval = [](const M3 &R) { return [R](V3 a, V3 b) { return std::acos(dot(R * a, b)); }; };
    return ransac_solve<2>(p1, p2, 5.99 * threshold * threshold, confidence, max_iteration, seed, solver, make_eval,
                           mask);
}

}   // namespace xrh
