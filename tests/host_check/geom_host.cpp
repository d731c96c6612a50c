// Test-only: exposes the PRODUCT's host geometry / config code (xrslam_amd/csrc/host/*.hpp) through a C
// interface so tests can check it against numpy.  Plain g++, no HIP.  Never shipped.
#include <cstring>

#include "../../xrslam_amd/csrc/host/config.hpp"
#include "../../xrslam_amd/csrc/host/geometry.hpp"

using namespace xrh;

extern "C" {

int gh_real_eigen(const double *M, int n, double *wr, double *wi, double *vecs /* n x n, column e = eigvec or zeros */) {
    Dense A(n, n);
    std::memcpy(A.a.data(), M, sizeof(double) * n * n);
    std::vector<double> r, i;
    std::vector<std::vector<double>> v;
    real_eigen(A, r, i, v);
    for (int e = 0; e < n; ++e) {
        wr[e] = r[e];
        wi[e] = i[e];
        for (int k = 0; k < n; ++k) vecs[k * n + e] = v[e].empty() ? 0.0 : v[e][k];
    }
    return 0;
}
int gh_svd(const double *M, int m, int n, double *s, double *V) {
    Dense A(m, n);
    std::memcpy(A.a.data(), M, sizeof(double) * m * n);
    std::vector<double> sv;
    Dense Vm;
    jacobi_svd(A, sv, Vm);
    for (int j = 0; j < n; ++j) s[j] = sv[j];
    std::memcpy(V, Vm.a.data(), sizeof(double) * n * n);
    return 0;
}
int gh_essential_5pt(const double *p1, const double *p2, double *E_out /* up to 10 x 9 */) {
    std::array<V2, 5> a, b;
    for (int i = 0; i < 5; ++i) {
        a[i] = {p1[2 * i], p1[2 * i + 1]};
        b[i] = {p2[2 * i], p2[2 * i + 1]};
    }
    auto Es = solve_essential_5pt(a, b);
    for (size_t k = 0; k < Es.size(); ++k) std::memcpy(E_out + 9 * k, Es[k].m, sizeof(double) * 9);
    return (int)Es.size();
}
void gh_rotation_2pt(const double *a6, const double *b6, double *R9) {
    std::array<V3, 2> a{V3{a6[0], a6[1], a6[2]}, V3{a6[3], a6[4], a6[5]}}, b{V3{b6[0], b6[1], b6[2]}, V3{b6[3], b6[4], b6[5]}};
    M3 R = solve_rotation_2pt(a, b);
    std::memcpy(R9, R.m, sizeof(R.m));
}
int gh_find_essential(const double *p1, const double *p2, int n, char *mask, double *E9) {
    std::vector<V2> a(n), b(n);
    for (int i = 0; i < n; ++i) {
        a[i] = {p1[2 * i], p1[2 * i + 1]};
        b[i] = {p2[2 * i], p2[2 * i + 1]};
    }
    std::vector<char> m;
    M3 E = find_essential_matrix(a, b, m, 1.0);
    std::memcpy(E9, E.m, sizeof(E.m));
    for (size_t i = 0; i < m.size(); ++i) mask[i] = m[i];
    return (int)m.size();
}
int gh_find_rotation(const double *p1, const double *p2, int n, double thr, char *mask, double *R9) {
    std::vector<V3> a(n), b(n);
    for (int i = 0; i < n; ++i) {
        a[i] = {p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]};
        b[i] = {p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]};
    }
    std::vector<char> m;
    M3 R = find_rotation_matrix(a, b, m, thr);
    std::memcpy(R9, R.m, sizeof(R.m));
    for (size_t i = 0; i < m.size(); ++i) mask[i] = m[i];
    return (int)m.size();
}
void gh_triangulate(const double *Ps, const double *zs, int n, double *h4) {
    std::vector<P34> P(n);
    std::vector<V3> z(n);
    for (int i = 0; i < n; ++i) {
        std::memcpy(P[i].m, Ps + 12 * i, sizeof(double) * 12);
        z[i] = {zs[3 * i], zs[3 * i + 1], zs[3 * i + 2]};
    }
    auto h = triangulate_point(P, z);
    for (int i = 0; i < 4; ++i) h4[i] = h[i];
}
void gh_lotbox(int size, unsigned seed, int rounds, int draws, long long *out) {
    LotBox box(size);
    box.seed(seed);
    for (int r = 0; r < rounds; ++r) {
        box.refill_all();
        for (int d = 0; d < draws; ++d) out[r * draws + d] = (long long)box.draw_without_replacement();
    }
}
int gh_load_config(const char *slam, const char *dev, double *out64) {
    try {
        Config c = load_config(slam, dev);
        double *o = out64;
        *o++ = c.cam_resolution[0]; *o++ = c.cam_resolution[1];
        *o++ = c.K.fx; *o++ = c.K.fy; *o++ = c.K.cx; *o++ = c.K.cy;
        *o++ = c.q_bc.x; *o++ = c.q_bc.y; *o++ = c.q_bc.z; *o++ = c.q_bc.w;
        *o++ = c.p_bc.x; *o++ = c.p_bc.y; *o++ = c.p_bc.z;
        *o++ = c.cov_g[0]; *o++ = c.cov_a[4]; *o++ = c.cov_bg[8]; *o++ = c.cov_ba[0];
        *o++ = (double)c.sliding_window_size; *o++ = (double)c.feature_tracker_max_keypoint_detection;
        *o++ = (double)c.feature_tracker_max_frames; *o++ = (double)c.solver_iteration_limit;
        *o++ = c.rotation_misalignment_threshold; *o++ = c.feature_tracker_predict_keypoints ? 1.0 : 0.0;
        *o++ = c.keypoint_noise_cov[0]; *o++ = (double)c.initializer_keyframe_num; *o++ = c.parsac_flag ? 1.0 : 0.0;
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}
}

// ------------------------------------------------------------------------------ two_view.hpp
#include "../../xrslam_amd/csrc/host/two_view.hpp"
extern "C" {
void gh_homography_4pt(const double *p1, const double *p2, double *H9) {
    std::array<V2, 4> a, b;
    for (int i = 0; i < 4; ++i) {
        a[i] = {p1[2 * i], p1[2 * i + 1]};
        b[i] = {p2[2 * i], p2[2 * i + 1]};
    }
    M3 H = solve_homography_4pt(a, b);
    std::memcpy(H9, H.m, sizeof(H.m));
}
int gh_find_homography(const double *p1, const double *p2, int n, double thr, int seed, char *mask, double *H9) {
    std::vector<V2> a(n), b(n);
    for (int i = 0; i < n; ++i) {
        a[i] = {p1[2 * i], p1[2 * i + 1]};
        b[i] = {p2[2 * i], p2[2 * i + 1]};
    }
    std::vector<char> m;
    M3 H = find_homography_matrix(a, b, m, thr, 0.999, 1000, seed);
    std::memcpy(H9, H.m, sizeof(H.m));
    int cnt = 0;
    for (size_t i = 0; i < m.size(); ++i) cnt += (mask[i] = m[i]);
    return cnt;
}
int gh_decompose_homography(const double *H9, double *R18, double *T6, double *n6) {
    M3 H, R1, R2;
    std::memcpy(H.m, H9, sizeof(H.m));
    V3 T1, T2, n1, n2;
    bool ok = decompose_homography(H, R1, R2, T1, T2, n1, n2);
    std::memcpy(R18, R1.m, sizeof(R1.m));
    std::memcpy(R18 + 9, R2.m, sizeof(R2.m));
    for (int k = 0; k < 3; ++k) {
        T6[k] = T1[k]; T6[3 + k] = T2[k];
        n6[k] = n1[k]; n6[3 + k] = n2[k];
    }
    return ok ? 1 : 0;
}
void gh_decompose_essential(const double *E9, double *R18, double *T3) {
    M3 E, R1, R2;
    std::memcpy(E.m, E9, sizeof(E.m));
    V3 T;
    decompose_essential(E, R1, R2, T);
    std::memcpy(R18, R1.m, sizeof(R1.m));
    std::memcpy(R18 + 9, R2.m, sizeof(R2.m));
    for (int k = 0; k < 3; ++k) T3[k] = T[k];
}
void gh_lstsq(const double *A, const double *b, int m, int n, double *x) {
    Dense D(m, n);
    std::memcpy(D.a.data(), A, sizeof(double) * m * n);
    std::vector<double> r = lstsq_qr(D, std::vector<double>(b, b + m));
    std::memcpy(x, r.data(), sizeof(double) * n);
}
void gh_svd_solve3(const double *A9, const double *b3, double *x3) {
    M3 A;
    std::memcpy(A.m, A9, sizeof(A.m));
    V3 x = svd_solve3(A, V3{b3[0], b3[1], b3[2]});
    for (int k = 0; k < 3; ++k) x3[k] = x[k];
}
void gh_quat_from_matrix(const double *R9, double *q4) {
    M3 R;
    std::memcpy(R.m, R9, sizeof(R.m));
    Quat q = quat_from_matrix(R);
    q4[0] = q.x; q4[1] = q.y; q4[2] = q.z; q4[3] = q.w;
}
void gh_from_two_vectors(const double *a3, const double *b3, double *q4) {
    Quat q = quat_from_two_vectors(V3{a3[0], a3[1], a3[2]}, V3{b3[0], b3[1], b3[2]});
    q4[0] = q.x; q4[1] = q.y; q4[2] = q.z; q4[3] = q.w;
}
void gh_logmap(const double *q4, double *w3) {
    V3 w = logmap(Quat{q4[0], q4[1], q4[2], q4[3]});
    for (int k = 0; k < 3; ++k) w3[k] = w[k];
}
void gh_s2_basis(const double *x3, double *b6) {
    V3 b1, b2;
    s2_tangential_basis(V3{x3[0], x3[1], x3[2]}, b1, b2);
    for (int k = 0; k < 3; ++k) {
        b6[k] = b1[k];
        b6[3 + k] = b2[k];
    }
}
}

// ------------------------------------------------------------------------------ parsac.hpp / epnp.hpp
#include "../../xrslam_amd/csrc/host/parsac.hpp"
extern "C" {
void gh_epnp(const double *X, const double *x, int n, double *R9, double *t3) {
    std::vector<V3> Xs(n);
    std::vector<V2> xs(n);
    for (int i = 0; i < n; ++i) {
        Xs[i] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
        xs[i] = {x[2 * i], x[2 * i + 1]};
    }
    Pose34 p = solve_pnp_epnp(Xs.data(), xs.data(), n);
    std::memcpy(R9, p.R.m, sizeof(p.R.m));
    for (int k = 0; k < 3; ++k) t3[k] = p.t[k];
}
// one persistent state per process, like the function-local statics of the reference
static ParsacState g_parsac;
void gh_parsac_reset() { g_parsac = ParsacState(); }
int gh_pnp_parsac_imu(const double *X, const double *x, const long *lens, int n, const double *R9, const double *t3, double dyn,
                      double thr, char *mask, double *Rout, double *tout) {
    std::vector<V3> Xs(n);
    std::vector<V2> xs(n);
    std::vector<size_t> ls(n);
    for (int i = 0; i < n; ++i) {
        Xs[i] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
        xs[i] = {x[2 * i], x[2 * i + 1]};
        ls[i] = (size_t)lens[i];
    }
    M3 R;
    std::memcpy(R.m, R9, sizeof(R.m));
    std::vector<char> m;
    Pose34 p = find_pnp_matrix_parsac_imu(g_parsac, Xs, xs, ls, R, V3{t3[0], t3[1], t3[2]}, dyn, 1.0, m, thr);
    int cnt = 0;
    for (size_t i = 0; i < m.size(); ++i) cnt += (mask[i] = m[i]);
    std::memcpy(Rout, p.R.m, sizeof(p.R.m));
    for (int k = 0; k < 3; ++k) tout[k] = p.t[k];
    return cnt;
}
int gh_essential_parsac(const double *p1, const double *p2, int n, double thr, char *mask, double *E9) {
    std::vector<V2> a(n), b(n);
    for (int i = 0; i < n; ++i) {
        a[i] = {p1[2 * i], p1[2 * i + 1]};
        b[i] = {p2[2 * i], p2[2 * i + 1]};
    }
    std::vector<char> m;
    M3 E = find_essential_matrix_parsac(g_parsac, a, b, m, thr);
    std::memcpy(E9, E.m, sizeof(E.m));
    int cnt = 0;
    for (size_t i = 0; i < m.size(); ++i) cnt += (mask[i] = m[i]);
    return cnt;
}
}

// PoissonDisk2 (csrc/host_select.hpp): the dense-grid fast path of permit() against the cell-by-cell scan with the hash fallback
// (a filter constructed without image bounds always takes the latter).  xy: n points, inserted in order into both; returns the
// number of points on which the two disagree (accepted by one, refused by the other).
#include "../../xrslam_amd/csrc/host_select.hpp"
extern "C" int gh_poisson_disagreements(const double *xy, int n, double radius, int w, int h, char *accepted) {
    xrhip::PoissonDisk2 dense(radius, w, h), sparse(radius);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const bool a = dense.insert(xy[2 * i], xy[2 * i + 1]);
        const bool b = sparse.insert(xy[2 * i], xy[2 * i + 1]);
        if (accepted) accepted[i] = a ? 1 : 0;
        bad += a != b;
    }
    return bad;
}
