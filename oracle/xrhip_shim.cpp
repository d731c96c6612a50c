// xrhip_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Implements the subset of the xrhip_* C ABI (include/xrslam_hip.h) that the host pipeline
// (xrslam_amd/csrc/host/pipeline.hpp) calls, on top of the CPU oracle.  Linking the pipeline sources
// against this shim instead of libxrslam_hip.so yields "the reference CPU path": identical host logic,
// oracle arithmetic, one thread.  Used by tests (full-pipeline parity: track ids / keypoint indices
// bit-exact, states within tolerance) and by bench.py's cpu_baseline leg.  The product never links it.
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../include/xrslam_hip.h"
#include "../xrslam_amd/csrc/hostprof.hpp"

extern "C" {
// oracle/klt_oracle.c
typedef struct OrcPyramid OrcPyramid;
OrcPyramid *orc_pyr_create(int w, int h, int max_level, int pad);
void orc_pyr_destroy(OrcPyramid *P);
void orc_preprocess(OrcPyramid *P, const uint8_t *img, int stride, double clip, int tx, int ty, uint8_t *work);
int orc_detect_keypoints(const uint8_t *img, int w, int h, int stride, const double *existing, int n_exist,
                         int max_points, double min_dist, double *out_xy);
void orc_track_keypoints(const OrcPyramid *A, const OrcPyramid *B, const double *curr, double *next_inout, int has_guess,
                         uint8_t *status, int n, void *stats);
void orc_set_threads(int n);
// oracle/ba_oracle.cpp
int orc_ba_solve(const xrhip_ba_problem *P, xrhip_ba_summary *summary);
int orc_ba_marginalize(const xrhip_marg_problem *M, double *out_sqrt_info, double *out_infovec, double *out_lin);
int orc_preintegrate(const double *samples, int n, double t_end, const double *bg, const double *ba,
                     const double *noise36, int jac, int cv, double *out);
}

struct xrhip_group;
struct xrhip_klt {
    xrhip_group *group = nullptr;
    int w, h;
    std::vector<uint32_t> undist_map;   // packed 1/32-pixel map, empty = frames arrive rectified
};
struct xrhip_image {
    xrhip_klt *ctx;
    std::vector<uint8_t> raw, clahe;
    OrcPyramid *pyr;
    bool have_raw, have_pyr;
};
struct xrhip_ba {
    xrhip_group *group = nullptr;
    const xrhip_ba_problem *begun_P = nullptr;   // xrhip_ba_solve_begin .. _end
    xrhip_ba_summary begun_sm;
    int begun_rc = 0;
    int unused = 0;
    // results of the asynchronous forms, per context like the product (computed at _begin, handed over at _end)
    std::vector<double> preint_out, marg_si, marg_iv, marg_lin;
    int preint_rc = 0, marg_rc = 0;
    int preint_pending = 0;   // jobs between _begin and _end (the product's state machine: ba_api.hip, preint_pending)
    // xrhip_ba_preintegrate_after_solve: the batch waits here for the next solve's biases
    struct Deferred {
        std::vector<double> samples, t_end, noise;
        std::vector<int> begin, count, frame;
        int jac = 0, cov = 0;
    };
    bool have_deferred = false;
    Deferred deferred;
};

static thread_local std::string g_err;

// Fault injection (tests/test_error_recovery.py): the `countdown`-th call from now of the entry point `which` fails with XRHIP_EHIP,
// the way a device error would surface there.  which: 1 = xrhip_ba_preintegrate_end, 2 = xrhip_ba_solve, 3 = xrhip_image_track,
// 4 = xrhip_ba_preintegrate_begin.  Process-wide (the tests drive one instance at a time).
#include <atomic>
static std::atomic<int> g_fail_which{0}, g_fail_countdown{0};
static bool injected_failure(int which, const char *name) {
    if (g_fail_which.load(std::memory_order_relaxed) != which) return false;
    if (g_fail_countdown.fetch_sub(1) != 1) return false;
    g_fail_which.store(0);
    g_err = std::string(name) + ": injected device failure";
    return true;
}
extern "C" void orc_inject_failure(int which, int countdown) {
    g_fail_countdown.store(countdown);
    g_fail_which.store(which);
}

// The checker's own solve / marginalisation clocks (bench.py's cpu_baseline leg: BASELINE.json's metric is "frames/sec + ms/BA-iteration";
// the reference times Solver::solve, estimation/solver.cpp:176-190).  Wall clock of orc_ba_solve / orc_ba_marginalize, process-wide,
// nanoseconds; out[0..4] = solve ms, solves, dogleg iterations, marginalisation ms, marginalisations.
#include <chrono>
static std::atomic<long long> g_clk_solve_ns{0}, g_clk_solves{0}, g_clk_iters{0}, g_clk_marg_ns{0}, g_clk_margs{0};
static inline long long clk_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
extern "C" void orc_shim_clocks(double *out5, int reset) {
    if (out5) {
        out5[0] = 1e-6 * (double)g_clk_solve_ns.load();
        out5[1] = (double)g_clk_solves.load();
        out5[2] = (double)g_clk_iters.load();
        out5[3] = 1e-6 * (double)g_clk_marg_ns.load();
        out5[4] = (double)g_clk_margs.load();
    }
    if (reset) {
        g_clk_solve_ns = 0;
        g_clk_solves = 0;
        g_clk_iters = 0;
        g_clk_marg_ns = 0;
        g_clk_margs = 0;
    }
}

extern "C" {

const char *xrhip_last_error(void) { return g_err.c_str(); }
int xrhip_device_count(void) { return 0; }
int xrhip_set_device(int) { return 0; }
int xrhip_get_device(int *device) {
    if (device) *device = 0;
    return 0;
}
int xrhip_bind_device(int) { return 0; }
const char *xrhip_kernel_revision(void) { return "cpu-oracle"; }

int xrhip_klt_create(int width, int height, int, xrhip_klt **out) {
    // XR_ORACLE_THREADS: threads of the image / point loops (OpenCV's parallel_for_ in the reference); the solver stays
    // single-threaded like the reference's num_threads = 1 (estimation/solver.cpp:185)
    if (const char *e = getenv("XR_ORACLE_THREADS")) orc_set_threads(atoi(e));
    *out = new xrhip_klt();
    (*out)->w = width;
    (*out)->h = height;
    return 0;
}
void xrhip_klt_destroy(xrhip_klt *c) {
    if (c && c->group) xrhip_klt_join_group(c, nullptr);
    delete c;
    xrhip::hostprof_dump();   // XRHIP_HOSTPROF=1: the host pipeline's named wall-clock accumulators, like the product library
}
int xrhip_image_create(xrhip_klt *c, xrhip_image **out) {
    xrhip_image *im = new xrhip_image();
    im->ctx = c;
    im->raw.resize((size_t)c->w * c->h);
    im->clahe.resize((size_t)c->w * c->h);
    im->pyr = orc_pyr_create(c->w, c->h, 3, 21);
    im->have_raw = im->have_pyr = false;
    *out = im;
    return 0;
}
void xrhip_image_destroy(xrhip_image *im) {
    if (!im) return;
    orc_pyr_destroy(im->pyr);
    delete im;
}
int xrhip_image_upload(xrhip_image *im, const uint8_t *gray, int stride) {
    for (int y = 0; y < im->ctx->h; ++y) std::memcpy(&im->raw[(size_t)y * im->ctx->w], gray + (size_t)y * stride, im->ctx->w);
    im->have_raw = true;
    im->have_pyr = false;
    return 0;
}
extern "C" void orc_remap_packed(const uint32_t *map2, int w, int h, const uint8_t *src, int sstride, uint8_t *dst, int dstride);
int xrhip_klt_set_undistort_map(xrhip_klt *c, const uint32_t *map2) {
    if (map2) c->undist_map.assign(map2, map2 + (size_t)2 * c->w * c->h);
    else c->undist_map.clear();
    return 0;
}
int xrhip_image_upload_distorted(xrhip_image *im, const void *gray, int stride, int) {
    if (im->ctx->undist_map.empty()) {
        g_err = "upload_distorted: no undistortion map";
        return XRHIP_ESTATE;
    }
    orc_remap_packed(im->ctx->undist_map.data(), im->ctx->w, im->ctx->h, static_cast<const uint8_t *>(gray), stride, im->raw.data(),
                     im->ctx->w);
    im->have_raw = true;
    im->have_pyr = false;
    return 0;
}
int xrhip_debug_get_raw(xrhip_image *im, uint8_t *out) {
    std::memcpy(out, im->raw.data(), im->raw.size());
    return 0;
}
int xrhip_image_upload_device(xrhip_image *im, const void *gray, int stride) {
    return xrhip_image_upload(im, static_cast<const uint8_t *>(gray), stride);   // "device" == host in the shim
}
int xrhip_image_preprocess(xrhip_image *im, double clip, int tx, int ty) {
    if (!im->have_raw) {
        g_err = "preprocess: no image";
        return XRHIP_ESTATE;
    }
    orc_preprocess(im->pyr, im->raw.data(), im->ctx->w, clip, tx, ty, im->clahe.data());
    im->have_pyr = true;
    return 0;
}
int xrhip_image_release(xrhip_image *im) {
    im->have_raw = im->have_pyr = false;
    return 0;
}
int xrhip_image_prefetch_detect(xrhip_image *) { return 0; }
int xrhip_image_detect(xrhip_image *im, const double *existing, int n_exist, int max_points, double min_dist,
                       double *out_xy, int *n_out) {
    if (!im->have_pyr) {
        g_err = "detect: preprocess() has not run";
        return XRHIP_ESTATE;
    }
    *n_out = orc_detect_keypoints(im->clahe.data(), im->ctx->w, im->ctx->h, im->ctx->w, existing, n_exist, max_points,
                                  min_dist, out_xy);
    return 0;
}
int xrhip_image_track_impl(const xrhip_image *cur, const xrhip_image *next, const double *curr_xy, double *next_xy,
                      int has_guess, uint8_t *status, int n) {
    if (!cur->have_pyr || !next->have_pyr) {
        g_err = "track: preprocess() has not run";
        return XRHIP_ESTATE;
    }
    if (n > 0) orc_track_keypoints(cur->pyr, next->pyr, curr_xy, next_xy, has_guess, status, n, nullptr);
    return 0;
}
int xrhip_image_track(const xrhip_image *cur, const xrhip_image *next, const double *curr_xy, double *next_xy, int has_guess,
                      uint8_t *status, int n) {
    if (injected_failure(3, "xrhip_image_track")) return XRHIP_EHIP;
    return xrhip_image_track_impl(cur, next, curr_xy, next_xy, has_guess, status, n);
}
int xrhip_klt_set_profiling(xrhip_klt *, int) { return 0; }
int xrhip_klt_get_stats(xrhip_klt *, xrhip_klt_stats *out, int) {
    std::memset(out, 0, sizeof(*out));
    return 0;
}
// instance groups: nothing to batch on the CPU -- a group is a membership count
struct xrhip_group {
    std::atomic<int> members{0};
};
int xrhip_group_create(xrhip_group **out) {
    *out = new xrhip_group();
    return 0;
}
int xrhip_group_destroy(xrhip_group *g) {
    if (g && g->members.load() != 0) {
        g_err = "xrhip_group_destroy: contexts are still joined to this group";
        return XRHIP_ESTATE;
    }
    delete g;
    return 0;
}
int xrhip_group_set_profiling(xrhip_group *, int) { return 0; }
int xrhip_group_queue_split(xrhip_group *) { return 0; }   // nothing is queued on the CPU build
int xrhip_group_get_stats(xrhip_group *, xrhip_group_stats *out, int) {
    if (out) std::memset(out, 0, sizeof(*out));
    return 0;
}
int xrhip_klt_frame_gate(xrhip_klt *) { return 0; }          // nothing is launched on the CPU: nothing to line up
int xrhip_klt_group_busy(xrhip_klt *, int) { return 0; }
int xrhip_klt_join_group(xrhip_klt *c, xrhip_group *g) {
    if (c->group) c->group->members.fetch_sub(1);
    c->group = g;
    if (g) g->members.fetch_add(1);
    return 0;
}
int xrhip_ba_join_group(xrhip_ba *c, xrhip_group *g) {
    if (c->group) c->group->members.fetch_sub(1);
    c->group = g;
    if (g) g->members.fetch_add(1);
    return 0;
}
int xrhip_ba_create(int, int, int, xrhip_ba **out) {
    *out = new xrhip_ba();
    return 0;
}
void xrhip_ba_destroy(xrhip_ba *c) {
    if (c && c->group) xrhip_ba_join_group(c, nullptr);
    delete c;
}
int xrhip_ba_preintegrate_begin(xrhip_ba *c, const double *samples, const int *begin, const int *count, const double *t_end,
                                const double *bg, const double *ba, int n_jobs, const double *noise36, int jac, int cov);
int xrhip_ba_solve(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *s);
int xrhip_ba_solve_overlapped(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *s, void (*host_work)(void *), void *arg) {
    const int rc = xrhip_ba_solve(c, P, s);   // nothing to overlap with on the CPU: the solve, then the caller's work
    if (host_work) host_work(arg);
    return rc;
}
int xrhip_ba_solve(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *s) {
    if (injected_failure(2, "xrhip_ba_solve")) {
        if (c) c->have_deferred = false;   // the product drops a batch staged behind a solve that failed
        return XRHIP_EHIP;
    }
    if (c && c->have_deferred && P)   // like the product: a bad frame index is refused before the solve touches anything
        for (int f : c->deferred.frame)
            if (f < 0 || f >= P->n_frames) {
                c->have_deferred = false;
                g_err = "xrhip_ba_preintegrate_after_solve: bias frame is not a frame of the solve";
                return XRHIP_EINVAL;
            }
    const long long t_solve = clk_ns();
    int rc = orc_ba_solve(P, s);
    {
        const long long dt = clk_ns() - t_solve;
        g_clk_solve_ns += dt;
        g_clk_solves += 1;
        if (rc == 0 && s) {
            g_clk_iters += s->iterations;
            s->ms_solve = 1e-6 * (double)dt;   // like the product's summary: wall clock of the whole solve
        }
    }
    if (rc == 0 && c && c->have_deferred) {   // the deferred batch starts from the biases this solve produced
        c->have_deferred = false;
        const xrhip_ba::Deferred &d = c->deferred;
        const int n = (int)d.frame.size();
        std::vector<double> bg(3 * (size_t)n), ba(3 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            if (d.frame[k] < 0 || d.frame[k] >= P->n_frames) {
                g_err = "xrhip_ba_preintegrate_after_solve: bias frame is not a frame of the solve";
                return XRHIP_EINVAL;
            }
            for (int i = 0; i < 3; ++i) {
                bg[3 * k + i] = P->frame_state[16 * d.frame[k] + 10 + i];
                ba[3 * k + i] = P->frame_state[16 * d.frame[k] + 13 + i];
            }
        }
        xrhip_ba_preintegrate_begin(c, d.samples.data(), d.begin.data(), d.count.data(), d.t_end.data(), bg.data(), ba.data(), n,
                                    d.noise.data(), d.jac, d.cov);
    }
    return rc;
}
// begin / linked / end: the CPU solves at begin (in place: the caller does not look before end) and hands the state over on the host
int xrhip_ba_solve_begin(xrhip_ba *c, const xrhip_ba_problem *P) {
    if (!c || !P) {
        g_err = "xrhip_ba_solve_begin: null argument";
        return XRHIP_EINVAL;
    }
    if (c->begun_P) {
        g_err = "xrhip_ba_solve_begin: a solve is already in flight on this context";
        return XRHIP_ESTATE;
    }
    // like the product: only the problems its single-launch kernel takes (no free landmark, no prior) are begun
    for (int l = 0; l < P->n_landmarks; ++l)
        if (!(P->landmark_fix && P->landmark_fix[l])) return 0;
    if (P->prior_n > 0 || c->have_deferred) return 0;
    c->begun_rc = xrhip_ba_solve(c, P, &c->begun_sm);
    c->begun_P = P;
    return 1;
}
int xrhip_ba_solve_end(xrhip_ba *c, xrhip_ba_summary *s) {
    if (!c || !c->begun_P) {
        g_err = "xrhip_ba_solve_end: nothing in flight";
        return XRHIP_ESTATE;
    }
    c->begun_P = nullptr;
    if (s) *s = c->begun_sm;
    return c->begun_rc;
}
int xrhip_ba_solve_abort(xrhip_ba *c) {
    if (c) c->begun_P = nullptr;
    return 0;
}
int xrhip_ba_solve_linked(xrhip_ba *c2, const xrhip_ba_problem *P2, xrhip_ba_summary *s2, int link_second, xrhip_ba *c1, int link_first,
                          void (*host_work)(void *), void *arg) {
    if (!c1 || !c2 || c1 == c2 || !P2 || !c1->begun_P || link_first < 0 || link_first >= c1->begun_P->n_frames || link_second < 0 ||
        link_second >= P2->n_frames) {
        g_err = "xrhip_ba_solve_linked: bad arguments / no solve in flight on the first context";
        return XRHIP_EINVAL;
    }
    if (c1->begun_rc) return c1->begun_rc;
    std::memcpy(P2->frame_state + 16 * (size_t)link_second, c1->begun_P->frame_state + 16 * (size_t)link_first, sizeof(double) * 16);
    return xrhip_ba_solve_overlapped(c2, P2, s2, host_work, arg);
}
int xrhip_ba_solve_chained(xrhip_ba *c1, const xrhip_ba_problem *P1, xrhip_ba_summary *s1, int link_first, xrhip_ba *c2,
                           const xrhip_ba_problem *P2, xrhip_ba_summary *s2, int link_second, void (*host_work)(void *), void *arg) {
    if (!c1 || !c2 || c1 == c2 || !P1 || !P2 || link_first < 0 || link_first >= P1->n_frames || link_second < 0 || link_second >= P2->n_frames) {
        g_err = "xrhip_ba_solve_chained: bad arguments";
        return XRHIP_EINVAL;
    }
    int rc = xrhip_ba_solve(c1, P1, s1);   // on the CPU: one after the other, the state handed over in between
    if (host_work) host_work(arg);
    if (rc) return rc;
    std::memcpy(P2->frame_state + 16 * (size_t)link_second, P1->frame_state + 16 * (size_t)link_first, sizeof(double) * 16);
    return xrhip_ba_solve(c2, P2, s2);
}
int xrhip_ba_debug_marg_guard(xrhip_ba *, double *, int *) {
    g_err = "xrhip_ba_debug_marg_guard: the CPU checker has no fast path to guard";
    return XRHIP_ESTATE;
}
int xrhip_ba_marginalize(xrhip_ba *, const xrhip_marg_problem *M, double *a, double *b, double *c) {
    const long long t_marg = clk_ns();
    int rc = orc_ba_marginalize(M, a, b, c);
    g_clk_marg_ns += clk_ns() - t_marg;
    g_clk_margs += 1;
    if (rc) g_err = "marginalize failed";
    return rc ? XRHIP_ESTATE : 0;
}
int xrhip_ba_preintegrate(xrhip_ba *, const double *samples, int n, double t_end, const double *bg, const double *ba,
                          const double *noise36, int jac, int cov, double *out) {
    int rc = orc_preintegrate(samples, n, t_end, bg, ba, noise36, jac, cov, out);
    if (rc) g_err = "preintegrate failed";
    return rc ? XRHIP_ESTATE : 0;
}
int xrhip_ba_set_profiling(xrhip_ba *, int) { return 0; }
int xrhip_ba_get_stats(xrhip_ba *, xrhip_ba_stats *out, int) {
    if (out) *out = xrhip_ba_stats{0, 0, 0.0, 0, 0.0, 0, 0, 0.0, 0.0};
    return 0;
}
int xrhip_ba_preintegrate_batch(xrhip_ba *c, const double *samples, const int *begin, const int *count,
                                const double *t_end, const double *bg, const double *ba, int n_jobs,
                                const double *noise36, int jac, int cov, double *out) {
    for (int k = 0; k < n_jobs; ++k) {
        int rc = xrhip_ba_preintegrate(c, samples + 7 * (size_t)begin[k], count[k], t_end[k], bg + 3 * k, ba + 3 * k,
                                       noise36, jac, cov, out + (size_t)XRHIP_IMU_DIM * k);
        if (rc) return rc;
    }
    return 0;
}
// asynchronous form: the CPU reference computes at _begin and hands the records over at _end
int xrhip_ba_preintegrate_begin(xrhip_ba *c, const double *samples, const int *begin, const int *count, const double *t_end,
                                const double *bg, const double *ba, int n_jobs, const double *noise36, int jac, int cov) {
    if (c->preint_pending || c->have_deferred) {   // the product's rule: one batch between _begin and _end per context
        g_err = "xrhip_ba_preintegrate_begin: a batch is already in flight on this context";
        return XRHIP_ESTATE;
    }
    if (injected_failure(4, "xrhip_ba_preintegrate_begin")) return XRHIP_EHIP;
    c->preint_out.assign((size_t)XRHIP_IMU_DIM * n_jobs, 0.0);
    c->preint_rc = xrhip_ba_preintegrate_batch(c, samples, begin, count, t_end, bg, ba, n_jobs, noise36, jac, cov, c->preint_out.data());
    c->preint_pending = n_jobs;
    return 0;
}
int xrhip_ba_preintegrate_after_solve(xrhip_ba *c, const double *samples, const int *begin, const int *count, const double *t_end,
                                      const int *bias_frame, int n_jobs, const double *noise36, int jac, int cov) {
    if (!c || !samples || !begin || !count || !t_end || !bias_frame || !noise36 || n_jobs <= 0) {
        g_err = "xrhip_ba_preintegrate_after_solve: bad arguments";
        return XRHIP_EINVAL;
    }
    if (c->preint_pending || c->have_deferred) {
        g_err = "xrhip_ba_preintegrate_after_solve: a batch is already in flight on this context";
        return XRHIP_ESTATE;
    }
    int total = 0;
    for (int k = 0; k < n_jobs; ++k) total = std::max(total, begin[k] + count[k]);
    xrhip_ba::Deferred &d = c->deferred;
    d.samples.assign(samples, samples + 7 * (size_t)total);
    d.begin.assign(begin, begin + n_jobs);
    d.count.assign(count, count + n_jobs);
    d.t_end.assign(t_end, t_end + n_jobs);
    d.frame.assign(bias_frame, bias_frame + n_jobs);
    d.noise.assign(noise36, noise36 + 36);
    d.jac = jac;
    d.cov = cov;
    c->have_deferred = true;
    return 0;
}
int xrhip_ba_preintegrate_cancel(xrhip_ba *c) {
    if (!c) return XRHIP_EINVAL;
    c->have_deferred = false;
    c->preint_rc = 0;
    c->preint_pending = 0;
    return 0;
}
int xrhip_ba_preintegrate_early(xrhip_ba *c, int job, double *out) {
    if (!c->preint_pending || job < 0 || job >= c->preint_pending) {
        g_err = "xrhip_ba_preintegrate_early: no such job in flight";
        return XRHIP_ESTATE;
    }
    if (c->preint_rc) return c->preint_rc;
    std::memcpy(out, c->preint_out.data() + (size_t)XRHIP_IMU_DIM * job, sizeof(double) * 11);
    return 0;
}
int xrhip_ba_preintegrate_end(xrhip_ba *c, double *out) {
    if (c->have_deferred) {
        c->have_deferred = false;
        g_err = "xrhip_ba_preintegrate_end: the batch waits for a solve that never ran";
        return XRHIP_ESTATE;
    }
    if (!c->preint_pending) {
        g_err = "xrhip_ba_preintegrate_end: nothing in flight";
        return XRHIP_ESTATE;
    }
    c->preint_pending = 0;
    if (injected_failure(1, "xrhip_ba_preintegrate_end")) return XRHIP_EHIP;
    if (c->preint_rc) return c->preint_rc;
    std::memcpy(out, c->preint_out.data(), sizeof(double) * c->preint_out.size());
    return 0;
}
// same for the marginalisation
int xrhip_ba_marginalize_begin(xrhip_ba *c, const xrhip_marg_problem *M) {
    const size_t R = 15 * (size_t)(M->n_frames - 1);
    c->marg_si.assign(R * R, 0.0);
    c->marg_iv.assign(R, 0.0);
    c->marg_lin.assign(16 * (size_t)(M->n_frames - 1), 0.0);
    c->marg_rc = xrhip_ba_marginalize(c, M, c->marg_si.data(), c->marg_iv.data(), c->marg_lin.data());
    return 0;
}
int xrhip_ba_marginalize_end(xrhip_ba *c, double *si, double *iv, double *lin) {
    if (c->marg_rc) return c->marg_rc;
    std::memcpy(si, c->marg_si.data(), sizeof(double) * c->marg_si.size());
    std::memcpy(iv, c->marg_iv.data(), sizeof(double) * c->marg_iv.size());
    std::memcpy(lin, c->marg_lin.data(), sizeof(double) * c->marg_lin.size());
    return 0;
}
}
