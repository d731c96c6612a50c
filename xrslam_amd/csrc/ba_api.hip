// ba_api.hip -- C ABI of the bundle adjustment (include/xrslam_hip.h, plug point #2).
// Host side of xrslam::Solver for gfx950: packs the factor graph into one device arena,
// builds the gather indices, and drives the kernel sequence of ba_kernels.hip.h.  The
// trust-region trial loop runs on the device (kb_solve_try); the host only intervenes when a new
// linearisation or a re-solve of the linear system is needed.
#include "../../include/xrslam_hip.h"
#include "ba_kernels.hip.h"
#include "ba_chain.hip.h"
#include "marg_kernels.hip.h"
#include "common.hip.h"
#include "group.hip.h"

#include <cstdlib>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace xrhip;

namespace {

struct Arena {   // bump allocator over one device buffer mirrored by a pinned host buffer
    char *dev = nullptr, *host = nullptr;
    size_t cap = 0, used = 0;
    size_t take(size_t bytes) {
        size_t off = (used + 255) & ~size_t(255);
        used = off + bytes;
        return off;
    }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// What a single-launch solve carries as a request: the staged problem's copy, the solve, and -- when an integration was queued
// behind the solve (xrhip_ba_preintegrate_after_solve) -- that batch, launched right behind it on the same stream
struct ChainPayload {
    StageArgs stage;
    ChainArgs chain;
    size_t lds = 0;   // (chain.link_src != null: a solve that starts from a state an earlier request's solve leaves on the device)
    bool with_preint = false;
    PreintArgs preint;
};
// One trust-region round of a window solve as a request of an instance group (GK_WROUND; ba_kernels.hip.h: WinEntry), or a batch of
// rejected trials (GK_WTRIALS: `e` alone).  A solve's first round also carries the copy of the staged problem and the prior's Lambda.
struct WindowPayload {
    WinEntry e;
    bool relin = false;        // this round linearises (kw_lin_all, kw_landmark_vision, kw_assemble)
    bool with_stage = false;   // first round of a solve
    StageArgs stage;
    int np = 0;                // > 0 with with_stage: kb_prior_lambda first
    gptr<const double> pS, pinfo;
    gptr<double> pLam, pc0;
    size_t lds_lin = 0, lds_solve = 0, lds_wide = 0;
};

}   // namespace

struct xrhip_ba {
    hipStream_t stream = nullptr;
    int device = 0;   // the device the context was created on (a group of another device is refused)
    // instance group (group.hip.h): the single-launch solves and the pre-integration batches travel as requests
    xrhip_group *group = nullptr;
    GroupRequest rq_chain, rq_preint, rq_window;
    WindowPayload a_window;
    ChainPayload a_chain;
    PreintArgs a_preint;
    GroupRequest *preint_rq = nullptr;   // the request that carries the batch in flight (grouped), and the stream it runs on
    hipStream_t preint_stream = nullptr;
    struct Begun {   // a single-launch solve queued by xrhip_ba_solve_begin, collected by xrhip_ba_solve_end
        bool active = false;
        const xrhip_ba_problem *P = nullptr;
        BaDims d{};
        BaPtrs p{};
        int seq = 0;
        hipStream_t stream = nullptr;
        GroupRequest *rq = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::chrono::steady_clock::time_point t0;
        // set when a linked solve took over (xrhip_ba_solve_linked): the summary's time runs to here -- the linked solve's own clock
        // covers the rest, so the two do not count the same wall time twice
        bool handed_over = false;
        std::chrono::steady_clock::time_point t_hand;
    } begun;
    Arena in;         // inputs (uploaded every solve)
    char *work = nullptr;   // device-only workspace
    size_t work_cap = 0;
    char *work2 = nullptr;  // marginalisation / pre-integration workspace
    size_t work2_cap = 0;
    char *h_stage = nullptr;   // pinned staging for work2 transfers
    size_t h_stage_cap = 0;
    BaCtl *h_ctl = nullptr;   // pinned; doubles as the zero-copy mailbox of kb_solve_try
    int *h_seq = nullptr;     // pinned; sequence number published by kb_solve_try after h_ctl / h_out
    int seq = 0;
    struct MargPending {   // a marginalisation queued by marg_launch and not yet collected
        bool pending = false;
        int K = 0, victim = 0, R = 0;
        std::vector<double> lin;
        int *dst = nullptr, *dsup = nullptr;
        double *lam = nullptr;   // km_chol's eigenvalue bound of the last marginalisation (xrhip_ba_debug_marg_guard)
        double *As = nullptr, *bs = nullptr, *B = nullptr, *V = nullptr, *Ss = nullptr, *ivs = nullptr, *dsi = nullptr, *div = nullptr;
    } marg;
    int preint_pending = 0;            // jobs of the pre-integration batch in flight (begin/end), 0 = none
    size_t preint_o_out = 0, preint_o_st = 0;
    // a batch staged by xrhip_ba_preintegrate_after_solve: launched by the next xrhip_ba_solve behind its last kernel
    int preint_deferred = 0, preint_def_jac = 0, preint_def_cov = 0;
    size_t preint_o_jobs = 0, preint_o_smp = 0, preint_o_noise = 0;
    // optional HIP-event profiling of kb_solve_try
    bool profiling = false;
    struct Timed {
        hipEvent_t e0, e1;
        double flops_fixed, flops_per_trial;
        int iter_before;
    };
    std::vector<Timed> pending;
    struct TimedChain {
        hipEvent_t e0, e1;
        double bytes;
    };
    std::vector<TimedChain> pending_chain;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    xrhip_ba_stats stats = {0, 0, 0.0, 0, 0.0, 0, 0, 0.0, 0.0};
    const TinyArgs *tiny_args = nullptr;   // device address of the staged argument block (kb_tiny)
    // speculative linearisation of window solves (spec_state_block): second stream, second set of linearisation buffers
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_mid = nullptr, ev_spec = nullptr;
    hipEvent_t ev_marg = nullptr;   // behind a marginalisation's last copy: collecting waits for THIS work, not for the stream
    char *work_spec = nullptr;
    size_t work_spec_cap = 0;
    bool spec_outstanding = false;         // stream2 may still be writing work_spec / reading the input arena
    long spec_launched = 0, spec_taken = 0;   // development counters (XRHIP_HOSTPROF prints them)
    const uint4 *stage_src = nullptr;      // the staged problem: pinned host block (device-visible address) -> device arena
    uint4 *stage_dst = nullptr;
    size_t stage_n16 = 0;
    void (*overlap_fn)(void *) = nullptr;   // xrhip_ba_solve_overlapped: pending host work of the caller (run at the first wait)
    void *overlap_arg = nullptr;
    double *h_out = nullptr;  // pinned readback (states + depths)
    size_t h_out_cap = 0;
    int lds_limit = 150 * 1024;
    int schur_mode = 0;                // study only (xrhip_ba_debug_set_schur_precision): 1 = f32, 2 = bf16 Schur contraction
    // last linearisation (debug/parity access)
    BaDims dims{};
    BaPtrs ptrs{};
    bool have_lin = false;
};

static int ensure_arena(xrhip_ba *c, size_t in_bytes, size_t work_bytes, size_t out_doubles) {
    if (in_bytes > c->in.cap) {
        if (c->in.dev) hipFree(c->in.dev);
        if (c->in.host) hipHostFree(c->in.host);
        c->in.dev = c->in.host = nullptr;
        size_t cap = std::max(in_bytes * 2, size_t(1) << 20);
        XR_HIP(hipMalloc(&c->in.dev, cap));
        XR_HIP(hipHostMalloc(&c->in.host, cap, hipHostMallocDefault));
        c->in.cap = cap;
    }
    if (work_bytes > c->work_cap) {
        if (c->work) hipFree(c->work);
        c->work = nullptr;
        size_t cap = std::max(work_bytes * 2, size_t(1) << 20);
        XR_HIP(hipMalloc(&c->work, cap));
        c->work_cap = cap;
    }
    if (out_doubles > c->h_out_cap) {
        if (c->h_out) hipHostFree(c->h_out);
        c->h_out = nullptr;
        XR_HIP(hipHostMalloc(&c->h_out, sizeof(double) * out_doubles * 2, hipHostMallocDefault));
        c->h_out_cap = out_doubles * 2;
    }
    return XRHIP_OK;
}

static int validate(const xrhip_ba_problem *P) {
    if (!P || P->n_frames <= 0 || !P->frame_state || !P->frame_fix) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: frames missing");
    if (P->n_landmarks < 0 || P->n_obs < 0 || P->n_rot < 0 || P->n_imu < 0 || P->prior_n < 0)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: negative count");
    for (int o = 0; o < P->n_obs; ++o)
        if (P->obs_tgt[o] < 0 || P->obs_tgt[o] >= P->n_frames || P->obs_ref[o] < 0 || P->obs_ref[o] >= P->n_frames ||
            P->obs_lm[o] < 0 || P->obs_lm[o] >= P->n_landmarks || P->obs_tgt[o] == P->obs_ref[o])
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: observation index out of range");
    for (int o = 0; o < P->n_rot; ++o)
        if (P->rot_tgt[o] < 0 || P->rot_tgt[o] >= P->n_frames || P->rot_ref[o] < 0 || P->rot_ref[o] >= P->n_frames)
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: rotation factor index out of range");
    for (int k = 0; k < P->n_imu; ++k)
        if (P->imu_i[k] < 0 || P->imu_i[k] >= P->n_frames || P->imu_j[k] < 0 || P->imu_j[k] >= P->n_frames ||
            P->imu_i[k] == P->imu_j[k])
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: imu factor index out of range");
    for (int k = 0; k < P->prior_n; ++k)
        if (P->prior_frames[k] < 0 || P->prior_frames[k] >= P->n_frames)
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: prior frame index out of range");
    return XRHIP_OK;
}

static int launch_stage_copy(xrhip_ba *c) {
    const int blocks = (int)std::min<size_t>((c->stage_n16 + 255) / 256, 128);
    Batch<StageArgs> b;
    std::memset(&b, 0, sizeof(b));
    b.e[0] = StageArgs{c->stage_src, c->stage_dst, c->stage_n16};
    hipLaunchKernelGGL(kb_stage, dim3(blocks, 1, 1), dim3(256), 0, c->stream, b);
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

// ---------------------------------------------------------------------------------------------- batched launches (group.hip.h)
static int launch_preint_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t = nullptr) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        Batch<PreintArgs> b;
        std::memset(&b, 0, sizeof(b));
        int most = 1;
        for (int i = 0; i < m; ++i) {
            b.e[i] = *static_cast<const PreintArgs *>(r[base + i]->payload);
            most = std::max(most, b.e[i].n_jobs);
        }
        hipLaunchKernelGGL(kp_preintegrate, dim3(most, 1, m), dim3(PI_NT), 0, s, b);
    }
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

static int launch_chain_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t = nullptr) {
    // In stream order: every staged problem is pulled from its pinned arena (16 bytes per lane, <= 128 workgroups per problem); one
    // workgroup solves each problem (dynamic LDS: the largest entry's layout) -- the solves that start from another request's
    // result (xrhip_ba_solve_linked) in a second launch, behind the first: the request they depend on is in this batch or an earlier
    // one (it was submitted first, and a queue launches in submission order); last the integrations that start from a solve's
    // biases read them where it left them.
    std::vector<StageArgs> stages;
    std::vector<std::pair<ChainArgs, size_t>> first, second;
    std::vector<PreintArgs> preints;
    for (int i = 0; i < n; ++i) {
        const ChainPayload &p = *static_cast<const ChainPayload *>(r[i]->payload);
        stages.push_back(p.stage);
        (p.chain.link_src ? second : first).emplace_back(p.chain, p.lds);   // a linked solve runs behind the one it starts from
        if (p.with_preint) preints.push_back(p.preint);
    }
    for (size_t base = 0; base < stages.size(); base += XB) {
        const int m = (int)std::min<size_t>(XB, stages.size() - base);
        Batch<StageArgs> bs;
        std::memset(&bs, 0, sizeof(bs));
        size_t most16 = 0;
        for (int i = 0; i < m; ++i) {
            bs.e[i] = stages[base + i];
            most16 = std::max(most16, bs.e[i].n16);
        }
        hipLaunchKernelGGL(kb_stage, dim3((int)std::min<size_t>((most16 + 255) / 256, 128), 1, m), dim3(256), 0, s, bs);
    }
    for (const auto *list : {&first, &second})
        for (size_t base = 0; base < list->size(); base += XB) {
            const int m = (int)std::min<size_t>(XB, list->size() - base);
            Batch<ChainArgs> bc;
            std::memset(&bc, 0, sizeof(bc));
            size_t lds = 0;
            for (int i = 0; i < m; ++i) {
                bc.e[i] = (*list)[base + i].first;
                lds = std::max(lds, (*list)[base + i].second);
            }
            hipLaunchKernelGGL(kb_chain, dim3(1, 1, m), dim3(CHAIN_THREADS), lds, s, bc);
        }
    for (size_t base = 0; base < preints.size(); base += XB) {
        const int m = (int)std::min<size_t>(XB, preints.size() - base);
        Batch<PreintArgs> bp;
        std::memset(&bp, 0, sizeof(bp));
        int most_jobs = 1;
        for (int i = 0; i < m; ++i) {
            bp.e[i] = preints[base + i];
            most_jobs = std::max(most_jobs, bp.e[i].n_jobs);
        }
        hipLaunchKernelGGL(kp_preintegrate, dim3(most_jobs, 1, m), dim3(PI_NT), 0, s, bp);
    }
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

// Window rounds of several members (round 5): every kernel of the round once, blockIdx.z = member.  In stream order: the staged
// problems of the solves that begin (kb_stage), their priors' Lambda (kb_prior_lambda, per entry: two launches in 27 frames), then the
// round -- entries that do not linearise (a re-solve at another trust-region radius) have grid size 0 in the first three kernels.
static int launch_wround_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t = nullptr) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        const WindowPayload *w[XB];
        for (int i = 0; i < m; ++i) w[i] = static_cast<const WindowPayload *>(r[base + i]->payload);
        Batch<StageArgs> bs;
        std::memset(&bs, 0, sizeof(bs));
        size_t most16 = 0;
        for (int i = 0; i < m; ++i)
            if (w[i]->with_stage) {
                bs.e[i] = w[i]->stage;
                most16 = std::max(most16, bs.e[i].n16);
            }
        if (most16) hipLaunchKernelGGL(kb_stage, dim3((int)std::min<size_t>((most16 + 255) / 256, 128), 1, m), dim3(256), 0, s, bs);
        for (int i = 0; i < m; ++i)
            if (w[i]->with_stage && w[i]->np) {
                const int np = w[i]->np;
                hipLaunchKernelGGL(kb_prior_lambda, dim3(((np + 15) / 16) * ((np + 15) / 16 + 1)), dim3(256), 0, s, np, w[i]->pS, w[i]->pLam, w[i]->pinfo,
                                   w[i]->pc0);
            }
        Batch<WinEntry> b;
        std::memset(&b, 0, sizeof(b));
        int g_lin = 0, g_lv = 0, g_asm = 0, g_sa = 0;
        size_t lds_lin = 0, lds_solve = 0, lds_wide = 0;
        for (int i = 0; i < m; ++i) {
            b.e[i] = w[i]->e;
            if (!w[i]->relin) b.e[i].g_lin = b.e[i].g_lv = b.e[i].g_asm = 0;
            g_lin = std::max(g_lin, b.e[i].g_lin);
            g_lv = std::max(g_lv, b.e[i].g_lv);
            g_asm = std::max(g_asm, b.e[i].g_asm);
            g_sa = std::max(g_sa, b.e[i].g_sa);
            lds_lin = std::max(lds_lin, w[i]->lds_lin);
            lds_solve = std::max(lds_solve, w[i]->lds_solve);
            lds_wide = std::max(lds_wide, w[i]->lds_wide);
        }
        if (g_lin) {
            hipLaunchKernelGGL(kw_lin_all, dim3(g_lin, 1, m), dim3(256), lds_lin, s, b);
            hipLaunchKernelGGL(kw_landmark_vision, dim3(std::max(g_lv, 1), 1, m), dim3(64), 0, s, b);
            hipLaunchKernelGGL(kw_assemble, dim3(std::max(g_asm, 1), 1, m), dim3(256), 0, s, b);
        }
        hipLaunchKernelGGL(kw_prepare, dim3(1, 1, m), dim3(256), 0, s, b);
        hipLaunchKernelGGL(kw_schur_aux, dim3(std::max(g_sa, 1), 1, m), dim3(256), 0, s, b);
        hipLaunchKernelGGL(kw_solve_try, dim3(1, 1, m), dim3(512), lds_solve, s, b);
        for (int i = 0; i < m; ++i) b.e[i].first = 1;
        hipLaunchKernelGGL(kw_trials_wide, dim3(WIDE_G, 1, m), dim3(256), lds_wide, s, b);
    }
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}
static int launch_wtrials_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t = nullptr) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        Batch<WinEntry> b;
        std::memset(&b, 0, sizeof(b));
        size_t lds_wide = 0;
        for (int i = 0; i < m; ++i) {
            const WindowPayload *w = static_cast<const WindowPayload *>(r[base + i]->payload);
            b.e[i] = w->e;
            b.e[i].first = 0;
            b.e[i].mode = 0;
            lds_wide = std::max(lds_wide, w->lds_wide);
        }
        hipLaunchKernelGGL(kw_trials_wide, dim3(WIDE_G, 1, m), dim3(256), lds_wide, s, b);
    }
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

namespace {
struct RegisterBaLaunchers {
    RegisterBaLaunchers() {
        group_register(GK_PREINT, launch_preint_batch);
        group_register(GK_CHAIN, launch_chain_batch);
        group_register(GK_WROUND, launch_wround_batch);
        group_register(GK_WTRIALS, launch_wtrials_batch);
    }
} g_register_ba_launchers;
}   // namespace

// Packs the problem + gather indices into the input arena, carves the workspace and fills dims/ptrs.
// defer_copy: do not queue the kb_stage launch; the caller hands the copy (c->stage_src / stage_dst / stage_n16) to a
// kernel that pulls the problem itself before it starts (kb_chain), or calls launch_stage_copy().
static int solve_lds(const BaDims &d, size_t limit, size_t *bytes, int *use_lds);
static int stage_problem(xrhip_ba *c, const xrhip_ba_problem *P, BaDims &d, BaPtrs &p, Ext &cam, Ext &imu, bool defer_copy = false) {
    d.F = P->n_frames;
    d.n = 15 * d.F;
    d.PF = round_up(6 * d.F, 16);
    d.L = P->n_landmarks;
    d.Lp = std::max(16, round_up(d.L, 16));
    d.M = P->n_obs;
    d.MR = P->n_rot;
    d.NI = P->n_imu;
    d.NP = P->prior_n;
    d.np = 15 * d.NP;
    d.NV = d.n + d.L;
    d.robust = 1;
    d.schur_mode = c->schur_mode;
    const int F = d.F, L = d.L, M = d.M, MR = d.MR, NI = d.NI, NP = d.NP, np = d.np, n = d.n;

    // ---- host-side index structures
    std::vector<uint8_t> lact(std::max(L, 1), 0);
    for (int o = 0; o < M; ++o) lact[P->obs_lm[o]] = 1;
    for (int l = 0; l < L; ++l)
        if (P->landmark_fix && P->landmark_fix[l]) lact[l] = 0;
    std::vector<int> lm_start(L + 1, 0), lm_obs(std::max(M, 1));
    for (int o = 0; o < M; ++o) lm_start[P->obs_lm[o] + 1]++;
    for (int l = 0; l < L; ++l) lm_start[l + 1] += lm_start[l];
    {
        std::vector<int> fill(lm_start.begin(), lm_start.end() - 1);
        for (int o = 0; o < M; ++o) lm_obs[fill[P->obs_lm[o]]++] = o;
    }
    std::vector<int> pair_start((size_t)F * F + 1, 0), pair_items(std::max(4 * M, 1));
    auto pair_count = [&](int a, int b) { pair_start[(size_t)a * F + b + 1]++; };
    for (int o = 0; o < M; ++o) {
        const int t = P->obs_tgt[o], r = P->obs_ref[o];
        pair_count(t, t);
        pair_count(r, r);
        pair_count(t, r);
        pair_count(r, t);
    }
    for (size_t i = 0; i < (size_t)F * F; ++i) pair_start[i + 1] += pair_start[i];
    {
        std::vector<int> fill(pair_start.begin(), pair_start.end() - 1);
        for (int o = 0; o < M; ++o) {
            const int t = P->obs_tgt[o], r = P->obs_ref[o];
            pair_items[fill[(size_t)t * F + t]++] = (o << 1) | 0;
            pair_items[fill[(size_t)r * F + r]++] = (o << 1) | 1;
            pair_items[fill[(size_t)t * F + r]++] = (o << 1) | 0;
            pair_items[fill[(size_t)r * F + t]++] = (o << 1) | 1;
        }
    }
    std::vector<int> rotf_start(F + 1, 0), rotf_items(std::max(MR, 1));
    for (int o = 0; o < MR; ++o) rotf_start[P->rot_tgt[o] + 1]++;
    for (int f = 0; f < F; ++f) rotf_start[f + 1] += rotf_start[f];
    {
        std::vector<int> fill(rotf_start.begin(), rotf_start.end() - 1);
        for (int o = 0; o < MR; ++o) rotf_items[fill[P->rot_tgt[o]]++] = o;
    }
    std::vector<int> act_idx;
    for (int a = 0; a < 15 * F; ++a) {
        const int f = a / 15, k = a % 15;
        if (k < 6 ? !(P->frame_fix[f] & XRHIP_FIX_POSE) : !(P->frame_fix[f] & XRHIP_FIX_MOTION)) act_idx.push_back(a);
    }
    d.na = (int)act_idx.size();
    std::vector<int> act_inv((size_t)15 * F, -1);
    for (size_t i = 0; i < act_idx.size(); ++i) act_inv[act_idx[i]] = (int)i;
    d.nla = 0;
    for (int l = 0; l < L; ++l) d.nla += lact[l] ? 1 : 0;
    d.lm_rows = d.Lp;   // xrhip_ba_solve drops the landmark rows when no landmark is free
    d.nfree = 0;
    for (int f = 0; f < F; ++f) d.nfree += (P->frame_fix[f] & 3) != 3 ? 1 : 0;
    d.nffp = 0;
    for (int o = 0; o < M; ++o)
        d.nffp += (!(P->frame_fix[P->obs_tgt[o]] & XRHIP_FIX_POSE) && !(P->frame_fix[P->obs_ref[o]] & XRHIP_FIX_POSE)) ? 1 : 0;
    std::vector<int> imuf(2 * F, -1), priorf(F, -1);
    for (int k = 0; k < NI; ++k) {
        if (imuf[2 * P->imu_j[k]] >= 0 || imuf[2 * P->imu_i[k] + 1] >= 0)
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: a frame may end at most one IMU factor and start at most one");
        imuf[2 * P->imu_j[k]] = k;
        imuf[2 * P->imu_i[k] + 1] = k;
    }
    for (int k = 0; k < NP; ++k) priorf[P->prior_frames[k]] = k;

    // ---- input arena layout
    Arena &A = c->in;
    A.used = 0;
    struct Item {
        size_t off;
        const void *src;
        size_t bytes;
    };
    std::vector<Item> items;
    auto put = [&](const void *src, size_t bytes) {
        size_t off = A.take(std::max(bytes, size_t(8)));
        items.push_back({off, src, bytes});
        return off;
    };
    std::vector<double> bias_ref((size_t)6 * std::max(NI, 1));
    for (int k = 0; k < NI; ++k)
        for (int i = 0; i < 6; ++i) bias_ref[6 * k + i] = P->frame_state[16 * P->imu_i[k] + 10 + i];
    BaCtl ctl;
    std::memset(&ctl, 0, sizeof(ctl));
    ctl.radius = 1e4;
    ctl.mu = 1e-8;
    ctl.first = 1;
    ctl.linear_ok = 1;
    ctl.max_iterations = P->max_iterations;
    ctl.termination = XRHIP_BA_NO_CONVERGENCE;
    ctl.accepted_slot = -1;
    const size_t o_state = put(P->frame_state, sizeof(double) * 16 * F);
    const size_t o_fix = put(P->frame_fix, F);
    const size_t o_depth = put(P->inv_depth, sizeof(double) * L);
    const size_t o_lact = put(lact.data(), std::max(L, 1));
    const size_t o_ot = put(P->obs_tgt, sizeof(int) * M), o_or = put(P->obs_ref, sizeof(int) * M);
    const size_t o_ol = put(P->obs_lm, sizeof(int) * M);
    const size_t o_zt = put(P->obs_z_tgt, sizeof(double) * 3 * M), o_zr = put(P->obs_z_ref, sizeof(double) * 3 * M);
    const size_t o_rt = put(P->rot_tgt, sizeof(int) * MR), o_rr = put(P->rot_ref, sizeof(int) * MR);
    const size_t o_rzt = put(P->rot_z_tgt, sizeof(double) * 3 * MR), o_rzr = put(P->rot_z_ref, sizeof(double) * 3 * MR);
    const size_t o_ii = put(P->imu_i, sizeof(int) * NI), o_ij = put(P->imu_j, sizeof(int) * NI);
    const size_t o_idata = put(P->imu_data, sizeof(double) * XRHIP_IMU_DIM * NI);
    const size_t o_bref = put(bias_ref.data(), sizeof(double) * 6 * NI);
    const size_t o_pf = put(P->prior_frames, sizeof(int) * NP);
    const size_t o_pS = put(P->prior_sqrt_info, sizeof(double) * (size_t)np * np);
    const size_t o_pi = put(P->prior_infovec, sizeof(double) * np);
    const size_t o_pl = put(P->prior_lin, sizeof(double) * 16 * NP);
    const size_t o_lms = put(lm_start.data(), sizeof(int) * (L + 1)), o_lmo = put(lm_obs.data(), sizeof(int) * M);
    const size_t o_ps = put(pair_start.data(), sizeof(int) * ((size_t)F * F + 1));
    const size_t o_pit = put(pair_items.data(), sizeof(int) * 4 * M);
    const size_t o_rfs = put(rotf_start.data(), sizeof(int) * (F + 1)), o_rfi = put(rotf_items.data(), sizeof(int) * MR);
    const size_t o_imuf = put(imuf.data(), sizeof(int) * 2 * F), o_prf = put(priorf.data(), sizeof(int) * F);
    const size_t o_act = put(act_idx.data(), sizeof(int) * act_idx.size());
    const size_t o_ainv = put(act_inv.data(), sizeof(int) * act_inv.size());
    const size_t o_ctl = put(&ctl, sizeof(ctl));
    const size_t o_args = put(nullptr, sizeof(TinyArgs));   // filled below, once the device addresses are known
    const size_t in_bytes = A.used + 256;

    // ---- workspace layout (device only)
    size_t w = 0;
    auto carve = [&](size_t bytes) {
        size_t off = (w + 255) & ~size_t(255);
        w = off + std::max(bytes, size_t(8));
        return off;
    };
    const size_t D8 = sizeof(double);
    const size_t w_cand = carve(D8 * 16 * F * TRY_B), w_dcand = carve(D8 * L * TRY_B);
    const size_t w_pLam = carve(D8 * (size_t)np * np), w_pc0 = carve(D8 * np);
    const size_t w_orec = carve(D8 * OREC * M), w_ocost = carve(D8 * M);
    const size_t w_rrec = carve(D8 * RREC * MR), w_rcost = carve(D8 * MR);
    const size_t w_ir = carve(D8 * 15 * NI), w_iJi = carve(D8 * 225 * NI), w_iJj = carve(D8 * 225 * NI);
    const size_t w_ic = carve(D8 * NI);
    const size_t w_pr = carve(D8 * np), w_pt = carve(D8 * np), w_pJq = carve(D8 * 9 * NP), w_pc = carve(D8 * (1 + (np + 15) / 16));
    const size_t w_H = carve(D8 * (size_t)n * n), w_g = carve(D8 * n);
    const size_t w_hll = carve(D8 * d.Lp), w_gl = carve(D8 * d.Lp), w_Wt = carve(D8 * (size_t)d.Lp * d.PF);
    const size_t w_sp = carve(D8 * n), w_sl = carve(D8 * d.Lp), w_om = carve(D8 * d.Lp);
    const size_t w_T = carve(D8 * (size_t)d.PF * d.PF), w_S = carve(D8 * (size_t)n * n);
    const size_t w_dD = carve(D8 * d.NV), w_gr = carve(D8 * d.NV), w_gn = carve(D8 * d.NV), w_gs = carve(D8 * d.NV);
    const size_t w_st = carve(D8 * d.NV), w_de = carve(D8 * d.NV * TRY_B), w_part = carve(D8 * (size_t)(aux_quad_blocks_n(d.n, std::max(d.L, 1)) + 8));
    const size_t w_wog = carve(D8 * d.PF * WOG_CH);
    const size_t w_wide = carve(D8 * WIDE_G * 4 * WIDE_B);
    const size_t w_Hv = carve(D8 * 36 * (size_t)F * F * VIS_CH), w_gv = carve(D8 * 6 * F * VIS_CH);
    int rc = ensure_arena(c, in_bytes, w + 256, (size_t)16 * F + L + 8);
    if (rc) return rc;
    for (const Item &it : items)
        if (it.bytes && it.src) std::memcpy(A.host + it.off, it.src, it.bytes);

    char *I = A.dev, *W = c->work;
    p.state = (double *)(I + o_state);
    p.cand = (double *)(W + w_cand);
    p.fix = (const uint8_t *)(I + o_fix);
    p.depth = (double *)(I + o_depth);
    p.depth_cand = (double *)(W + w_dcand);
    p.lact = (const uint8_t *)(I + o_lact);
    p.obs_tgt = (const int *)(I + o_ot);
    p.obs_ref = (const int *)(I + o_or);
    p.obs_lm = (const int *)(I + o_ol);
    p.obs_zt = (const double *)(I + o_zt);
    p.obs_zr = (const double *)(I + o_zr);
    p.rot_tgt = (const int *)(I + o_rt);
    p.rot_ref = (const int *)(I + o_rr);
    p.rot_zt = (const double *)(I + o_rzt);
    p.rot_zr = (const double *)(I + o_rzr);
    p.imu_i = (const int *)(I + o_ii);
    p.imu_j = (const int *)(I + o_ij);
    p.imu_data = (const double *)(I + o_idata);
    p.bias_ref = (double *)(I + o_bref);
    p.prior_frames = (const int *)(I + o_pf);
    p.pS = (const double *)(I + o_pS);
    p.pinfo = (const double *)(I + o_pi);
    p.plin = (const double *)(I + o_pl);
    p.pLam = (double *)(W + w_pLam);
    p.pc0 = (double *)(W + w_pc0);
    p.lm_start = (const int *)(I + o_lms);
    p.lm_obs = (const int *)(I + o_lmo);
    p.pair_start = (const int *)(I + o_ps);
    p.pair_items = (const int *)(I + o_pit);
    p.rotf_start = (const int *)(I + o_rfs);
    p.rotf_items = (const int *)(I + o_rfi);
    p.imuf = (const int *)(I + o_imuf);
    p.priorf = (const int *)(I + o_prf);
    p.act_idx = (const int *)(I + o_act);
    p.act_inv = (const int *)(I + o_ainv);
    p.Hv = (double *)(W + w_Hv);
    p.gv = (double *)(W + w_gv);
    p.orec = (double *)(W + w_orec);
    p.ocost = (double *)(W + w_ocost);
    p.rrec = (double *)(W + w_rrec);
    p.rcost = (double *)(W + w_rcost);
    p.imu_r = (double *)(W + w_ir);
    p.imu_Ji = (double *)(W + w_iJi);
    p.imu_Jj = (double *)(W + w_iJj);
    p.imu_cost = (double *)(W + w_ic);
    p.pr = (double *)(W + w_pr);
    p.pt = (double *)(W + w_pt);
    p.pJq = (double *)(W + w_pJq);
    p.pcost = (double *)(W + w_pc);
    p.Hpp = (double *)(W + w_H);
    p.gp = (double *)(W + w_g);
    p.hll = (double *)(W + w_hll);
    p.gl = (double *)(W + w_gl);
    p.Wt = (double *)(W + w_Wt);
    p.sp = (double *)(W + w_sp);
    p.sl = (double *)(W + w_sl);
    p.omega = (double *)(W + w_om);
    p.T = (double *)(W + w_T);
    p.Sred = (double *)(W + w_S);
    p.diagD = (double *)(W + w_dD);
    p.grad = (double *)(W + w_gr);
    p.gn = (double *)(W + w_gn);
    p.gs = (double *)(W + w_gs);
    p.step = (double *)(W + w_st);
    p.delta = (double *)(W + w_de);
    p.partial = (double *)(W + w_part);
    p.wog = (double *)(W + w_wog);
    p.wide_part = (double *)(W + w_wide);
    p.ctl = (BaCtl *)(I + o_ctl);
    XR_HIP(hipHostGetDevicePointer((void **)&p.host_ctl, c->h_ctl, 0));
    XR_HIP(hipHostGetDevicePointer((void **)&p.host_out, c->h_out, 0));
    XR_HIP(hipHostGetDevicePointer((void **)&p.host_seq, c->h_seq, 0));
    cam.q = Q4{P->cam_q_bc[0], P->cam_q_bc[1], P->cam_q_bc[2], P->cam_q_bc[3]};
    cam.p = V3{P->cam_p_bc[0], P->cam_p_bc[1], P->cam_p_bc[2]};
    imu.q = Q4{P->imu_q_bi[0], P->imu_q_bi[1], P->imu_q_bi[2], P->imu_q_bi[3]};
    imu.p = V3{P->imu_p_bi[0], P->imu_p_bi[1], P->imu_p_bi[2]};
    {   // a reduced system that does not fit LDS is written (kb_schur_aux) and factored (kb_solve_try) in the tiled layout, in place
        size_t lds_b = 0;
        int use_lds_ = 1;
        d.sred_tiled = 0;
        solve_lds(d, (size_t)c->lds_limit, &lds_b, &use_lds_);
        static const bool no_tiled = std::getenv("XRHIP_NO_TILED") != nullptr;
        d.sred_tiled = (use_lds_ == 0 && !no_tiled) ? 1 : 0;
    }
    {   // the argument block of the single-launch solve travels with the problem
        TinyArgs *ta = reinterpret_cast<TinyArgs *>(A.host + o_args);
        ta->d = d;
        ta->p = p;
        ta->cam = cam;
        ta->imu = imu;
        ta->sx = P->sqrt_inv_cov[0];
        ta->sy = P->sqrt_inv_cov[1];
        c->tiny_args = reinterpret_cast<const TinyArgs *>(I + o_args);
    }
    {
        char *host_dev = nullptr;   // device-visible address of the pinned staging block
        XR_HIP(hipHostGetDevicePointer((void **)&host_dev, A.host, 0));
        c->stage_src = (const uint4 *)host_dev;
        c->stage_dst = (uint4 *)A.dev;
        c->stage_n16 = (in_bytes + 15) / 16;
        if (!defer_copy) return launch_stage_copy(c);
    }
    return XRHIP_OK;
}

// small problems without a free landmark: assembly, preparation and the reduced system in one workgroup (kb_small_mid)
// no free landmark and a single free frame (localize_newframe, the initialiser's PnP): the whole solve runs inside
// one launch (kb_tiny).  Measured: with several free frames (refine_subwindow, na = 30..60) the one-workgroup
// assembly of the active block (26 of its 41 us: ~60 dependent-latency loads per entry, 8 entries per thread) costs more
// than the launches and round trips it saves (0.243 vs 0.215 ms per frame; index tables in LDS did not change that).
static bool tiny(const BaDims &d) {
    static const bool off = std::getenv("XRHIP_NO_TINY") != nullptr;   // development switch: force the multi-launch path
    return !off && d.nla == 0 && d.na <= 16 && d.M + d.MR <= 640 && d.F <= 64;   // kb_tiny lists the free frames in s_free[64]
}
static bool small_mid(const BaDims &d) { return d.nla == 0 && d.na <= 16; }   // measured: beyond one free frame the wide launches win
// no free landmark, no prior, a handful of free frames: the LDS-resident single-launch solve (ba_chain.hip.h)
static bool chain(const BaDims &d, size_t lds_limit, size_t *lds_bytes, int *vis_tile) {
    static const bool off = std::getenv("XRHIP_NO_CHAIN") != nullptr;   // development switch: the round-1 paths
    if (off || d.nla != 0 || d.NP != 0 || d.nffp != 0) return false;
    if (d.na < 1 || d.na > CHAIN_MAX_NA || d.NI > CHAIN_MAX_NI || d.F > CHAIN_MAX_F || d.M + d.MR > CHAIN_MAX_OBS ||
        d.nfree > CHAIN_MAX_FREE)
        return false;
    const size_t bytes = sizeof(double) * (size_t)chain_layout(d.F, d.na, d.NI, d.nfree, d.M + d.MR).total;
    if (bytes > lds_limit || d.F > 255) return false;
    // optional regions behind the layout, while they fit: the reduction tile of the reprojection blocks, then the table of
    // per-solve constants of the reprojection factors (the bigger win: taken first when only one fits)
    const size_t tile = sizeof(double) * (size_t)CHAIN_VIS_TILE;
    const size_t cache = sizeof(double) * (size_t)CHAIN_OBS_CACHE * (size_t)chain_cache_stride(d.M);
    static const bool no_cache = std::getenv("XRHIP_NO_OBS_CACHE") != nullptr;   // development switch (A/B, parity)
    int o = 0;
    size_t total = bytes;
    if (!no_cache && d.M > 0 && total + cache <= lds_limit) {
        o |= CHAIN_OPT_CACHE;
        total += cache;
    }
    if (total + tile <= lds_limit) {
        o |= CHAIN_OPT_TILE;
        total += tile;
    }
    *vis_tile = o;
    *lds_bytes = total;
    return true;
}

// one linearisation: 3 launches + the cost (and, for the solver, gradient norm + per-solve preparation)
static void launch_linearize(xrhip_ba *c, const BaDims &d, const BaPtrs &p, const Ext &cam, const Ext &imu, double sx,
                             double sy, bool for_solver, hipStream_t s_override = nullptr) {
    hipStream_t s = s_override ? s_override : c->stream;
    hipLaunchKernelGGL(kb_lin_all, dim3(lin_all_blocks(d.M, d.MR, d.NI, d.np)), dim3(256), sizeof(double) * std::max(d.np, 1), s, d, p,
                       cam, imu, sx, sy);
    hipLaunchKernelGGL(kb_landmark_vision, dim3(d.lm_rows + d.F * d.F * VIS_CH), dim3(64), 0, s, d, p);
    if (for_solver && small_mid(d)) return;   // kb_small_mid (launch_solve_try) assembles what the solve reads
    hipLaunchKernelGGL(kb_assemble, dim3((d.n * d.n + 255) / 256), dim3(256), 0, s, d, p);
    if (for_solver) hipLaunchKernelGGL(kb_prepare, dim3(1), dim3(256), 0, s, d, p);   // (cost + gradient norm: the last block of kb_schur_aux)
    else hipLaunchKernelGGL(kb_sum_cost, dim3(1), dim3(256), 0, s, d, p);
}

// dynamic LDS of solve_block: work region = max(packed triangle incl. the rhs row that rides along, gathered
// frame step of the back-substitution); systems whose triangle does not fit `limit` are factored in the global
// buffer Sred instead
static int solve_lds(const BaDims &d, size_t limit, size_t *bytes, int *use_lds) {
    static const bool no_tiled = std::getenv("XRHIP_NO_TILED") != nullptr;   // development switch (A/B, parity)
    const size_t tri = (size_t)(d.na + 1) * (d.na + 2) / 2;
    const size_t tiled = (size_t)tl_doubles(d.na + 1) + 16 * (size_t)tl_tile_rows(d.na + 1);   // tiles + L^-1 rhs
    const size_t aux = (size_t)d.PF;
    size_t lds = sizeof(double) * std::max(tiled, aux);
    *use_lds = 2;   // tiled layout (dense_lds.hip.h, round 3)
    if (no_tiled || lds > limit) {
        lds = sizeof(double) * std::max(tri, aux);
        *use_lds = 1;   // packed triangle in LDS
    }
    if (lds > limit) {
        *use_lds = 0;   // factored in the global buffer Sred, in the tiled layout (BaDims::sred_tiled); L^-1 rhs / the solution in LDS
        lds = sizeof(double) * std::max(aux, (size_t)16 * tl_tile_rows(d.na + 1));
    }
    *bytes = lds;
    return XRHIP_OK;
}

// Large problems hand a run of rejected trials to kb_trials_wide (the whole chip costs 8 candidates per launch);
// for small ones the round trip would cost more than looping inside kb_solve_try.
static bool wide_trials(const BaDims &d) { return d.M >= 256 && d.F <= 32; }
// ... and window-sized problems (refine_window) cost even their first trial there, queued right behind the solve
static bool wide_first(const BaDims &d) { return wide_trials(d) && d.M >= 600 && d.na >= 90; }

// reduced-system solve + trust-region trials: 2 launches (3 when mu changed without a new linearisation)
static void launch_schur_aux(const BaDims &d, const BaPtrs &p, hipStream_t s) {
    const int tiles = d.PF / 16;
    const int nrest = sred_rest_blocks(d);   // blocks that write the Schur-free entries of the reduced system (+ the tiled layout's padding)
    hipLaunchKernelGGL(kb_schur_aux, dim3(nrest + (d.nla ? tiles * tiles + aux_quad_blocks_n(d.n, d.L) + aux_wog_blocks(d.F)
                                                        : aux_quad_blocks_n(d.n, d.L)) + 1),   // + 1: total cost and gradient max-norm
                       dim3(256), 0, s, d, p);
}

// skip_front: the reduced system is already there (a committed speculation built it on the second stream)
static int launch_solve_try(xrhip_ba *c, const BaDims &d, const BaPtrs &p, const Ext &cam, const Ext &imu, double sx,
                            double sy, bool prepare, int mode, int seq, bool skip_front = false, const BaPtrs *spec_set = nullptr) {
    // spec_set: speculative linearisation is on.  The factorisation kernel leaves the first candidate in the second buffer
    // set; the trials are costed on the SECOND stream (they pay the cross-stream hand-over, ~14 us, off the critical path),
    // because the caller queues the linearisation chain at the candidate on THIS stream right behind the factorisation --
    // and, when the candidate is accepted, the next factorisation right behind that chain, without any event in between.
    // skip_front (with spec_set): that accepted case -- the kernel first takes cost / gradient norm over from the chain.
    const bool mark_mid = spec_set != nullptr;
    const int commit = (spec_set && skip_front) ? 1 : 0;
    double *state2 = spec_set ? static_cast<double *>(spec_set->state) : nullptr;
    double *depth2 = spec_set ? static_cast<double *>(spec_set->depth) : nullptr;
    BaCtl *ctl2 = spec_set ? static_cast<BaCtl *>(spec_set->ctl) : nullptr;
    hipStream_t s = c->stream;
    const bool mid = small_mid(d);
    if (!skip_front) {
        if (mid) hipLaunchKernelGGL(kb_small_mid, dim3(1), dim3(256), 0, s, d, p, prepare ? 0 : 1);
        if (!mid && prepare) hipLaunchKernelGGL(kb_prepare, dim3(1), dim3(256), 0, s, d, p);
        if (!mid) launch_schur_aux(d, p, s);
    }
    size_t lds = 0;
    int use_lds = 1;
    int rcl = solve_lds(d, (size_t)c->lds_limit, &lds, &use_lds);
    if (rcl) return rcl;
    lds = std::max(lds, sizeof(double) * (size_t)std::max(TRY_B * (d.np + 15 * d.NI), 1));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        if (!c->free_events.empty()) {
            e0 = c->free_events.back().first;
            e1 = c->free_events.back().second;
            c->free_events.pop_back();
        } else {
            XR_HIP(hipEventCreate(&e0));
            XR_HIP(hipEventCreate(&e1));
        }
        XR_HIP(hipEventRecord(e0, s));
    }
    if (wide_first(d))
        hipLaunchKernelGGL((kb_solve_try<512, true>), dim3(1), dim3(512), lds, s, d, p, cam, imu, sx, sy, use_lds, mode, seq, 2, state2, depth2, ctl2, commit);
    else if (d.M + d.MR <= 640 && d.na <= 64)
        hipLaunchKernelGGL((kb_solve_try<256, false>), dim3(1), dim3(256), lds, s, d, p, cam, imu, sx, sy, use_lds, mode, seq,
                           wide_trials(d) ? 1 : 0, (double *)nullptr, (double *)nullptr, (BaCtl *)nullptr, 0);
    else
        hipLaunchKernelGGL((kb_solve_try<512, false>), dim3(1), dim3(512), lds, s, d, p, cam, imu, sx, sy, use_lds, mode, seq,
                           wide_trials(d) ? 1 : 0, (double *)nullptr, (double *)nullptr, (BaCtl *)nullptr, 0);
    XR_HIP(hipGetLastError());
    if (c->profiling) XR_HIP(hipEventRecord(e1, s));   // the events bracket kb_solve_try alone (what rocprofv3 reports for it)
    if (wide_first(d)) {   // the first trial batch rides right behind the solve: no host round trip in between
        hipStream_t st = s;
        if (mark_mid) {
            XR_HIP(hipEventRecord(c->ev_mid, s));
            XR_HIP(hipStreamWaitEvent(c->stream2, c->ev_mid, 0));
            st = c->stream2;
        }
        const size_t wlds = sizeof(double) * ((size_t)WIDE_B * (16 * (size_t)d.F + (size_t)d.np) + (size_t)4 * WIDE_B * 257);
        hipLaunchKernelGGL(kb_trials_wide, dim3(WIDE_G), dim3(256), wlds, st, d, p, cam, imu, sx, sy, seq, 1, mode);
        XR_HIP(hipGetLastError());
    }
    if (c->profiling) {
        const double na = d.na;
        // algorithmic flops of this kernel: the factorisation + substitutions, and -- unless the trials are costed by
        // kb_trials_wide -- the factor evaluations of every trial it runs
        c->pending.push_back({e0, e1, na * na * na / 3.0 + 2.0 * na * na,
                              wide_first(d) ? 0.0 : 450.0 * d.M + 3000.0 * d.NI + 2.0 * (double)d.np * d.np, -1});
    }
    c->stats.n_solve_try++;
    return XRHIP_OK;
}

// Spin on the sequence number kb_solve_try stores (system-scope release) after its results: completion is seen a
// few microseconds after the kernel's last store, where a blocking hipStreamSynchronize costs 20-30 us.
// hipStreamQuery is polled now and then so that a faulted kernel turns into an error instead of a hang.
// xrhip_ba_solve_overlapped: the caller's host work runs here, once, at the solve's first wait
static void run_overlap(xrhip_ba *c) {
    if (!c->overlap_fn) return;
    void (*fn)(void *) = c->overlap_fn;
    c->overlap_fn = nullptr;
    fn(c->overlap_arg);
}
static int wait_mailbox(xrhip_ba *c, int seq, hipStream_t publisher = nullptr, GroupRequest *rq = nullptr) {
    run_overlap(c);
    return wait_flag(c->h_seq, seq, publisher ? publisher : c->stream, rq, "xrhip_ba_solve");
}

static long long g_kprof[32];   // accumulated in-kernel phase ticks (all zero unless built with -DXRHIP_KPROF)
// development aid: XRHIP_KPROF_MIN_NA / _MAX_NA restrict the accumulated phase timers to solves of that many active unknowns
static void kprof_accumulate(const BaCtl &ctl, int na) {
    static const int min_na = std::getenv("XRHIP_KPROF_MIN_NA") ? std::atoi(std::getenv("XRHIP_KPROF_MIN_NA")) : 0;
    static const int max_na = std::getenv("XRHIP_KPROF_MAX_NA") ? std::atoi(std::getenv("XRHIP_KPROF_MAX_NA")) : 1 << 30;
    if (na < min_na || na > max_na) return;
    for (int i = 0; i < 31; ++i) g_kprof[i] += ctl.prof[i];
    g_kprof[31] += 1;   // solves counted
}

extern "C" {

static void ba_resolve_pending(xrhip_ba *c) {
    for (auto &t : c->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            c->stats.ms_solve_try += ms;
            c->stats.n_timed += 1;
            c->stats.flops_solve_try += t.flops_fixed + t.flops_per_trial * std::max(t.iter_before, 0);
        }
        c->free_events.push_back({t.e0, t.e1});
    }
    c->pending.clear();
    for (auto &t : c->pending_chain) {
        float ms = 0.f;
        if (hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            c->stats.ms_chain += ms;
            c->stats.n_chain_timed += 1;
            c->stats.bytes_chain += t.bytes;
        }
        c->free_events.push_back({t.e0, t.e1});
    }
    c->pending_chain.clear();
}

int xrhip_ba_set_profiling(xrhip_ba *c, int enable) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_set_profiling: null context");
    if (!enable) ba_resolve_pending(c);
    c->profiling = enable != 0;
    return XRHIP_OK;
}

int xrhip_ba_get_stats(xrhip_ba *c, xrhip_ba_stats *out, int reset) {
    if (!c || !out) return xr_fail(XRHIP_EINVAL, "xrhip_ba_get_stats: null argument");
    ba_resolve_pending(c);
    *out = c->stats;
    if (reset) c->stats = {0, 0, 0.0, 0, 0.0, 0, 0, 0.0, 0.0};
    return XRHIP_OK;
}

/* development aid: in-kernel phase timers of kb_solve_try accumulated over all solves of the process,
 * in 100 MHz ticks; only instrumented builds (build.sh -DXRHIP_KPROF) write them */
void xrhip_debug_kprof(long long *out32, int reset) {
    for (int i = 0; i < 32; ++i) {
        if (out32) out32[i] = g_kprof[i];
        if (reset) g_kprof[i] = 0;
    }
}

int xrhip_ba_create(int max_frames, int max_landmarks, int max_obs, xrhip_ba **out) {
    if (!out || max_frames <= 0 || max_landmarks < 0 || max_obs < 0) return xr_fail(XRHIP_EINVAL, "xrhip_ba_create: bad arguments");
    int rc = xr_require_device();
    if (rc) return rc;
    xrhip_ba *c = new xrhip_ba();
    XR_HIP(hipGetDevice(&c->device));
    XR_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    XR_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    XR_HIP(hipEventCreateWithFlags(&c->ev_mid, hipEventDisableTiming));
    XR_HIP(hipEventCreateWithFlags(&c->ev_spec, hipEventDisableTiming));
    XR_HIP(hipHostMalloc(&c->h_ctl, sizeof(BaCtl), hipHostMallocDefault));
    XR_HIP(hipHostMalloc(&c->h_seq, 64, hipHostMallocDefault));
    *c->h_seq = 0;
    XR_HIP(hipFuncSetAttribute((const void *)kb_solve_try<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)kb_solve_try<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)kb_solve_try<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)kb_chain, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)kb_trials_wide, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)kw_solve_try, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)kw_trials_wide, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)km_chol, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    XR_HIP(hipFuncSetAttribute((const void *)km_jacobi, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit));
    // pre-size for the advertised maxima
    size_t in_guess = (size_t)max_obs * 96 + (size_t)max_frames * max_frames * 15 * 15 * 8 + (size_t)max_frames * 3000 + 65536;
    rc = ensure_arena(c, in_guess, in_guess * 4, (size_t)16 * max_frames + max_landmarks + 8);
    if (rc) return rc;
    *out = c;
    return XRHIP_OK;
}

int xrhip_ba_join_group(xrhip_ba *c, xrhip_group *g) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_join_group: null context");
    if (c->group == g) return XRHIP_OK;
    if (g && group_device(g) != c->device)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_join_group: the context and the group live on different devices");
    if (c->preint_pending) {   // a batch between begin and end: it completes where it was queued and stays collectable
        if (c->preint_rq) {
            int rc = group_wait_launched(c->preint_rq);
            if (rc) return rc;
        }
        XR_HIP(hipStreamSynchronize(c->preint_stream ? c->preint_stream : c->stream));
        c->preint_rq = nullptr;
        c->preint_stream = c->stream;
    }
    XR_HIP(hipStreamSynchronize(c->stream));
    if (c->group) {
        int rc = group_drain(c->group, GQ_CHAIN, c);
        if (!rc) rc = group_drain(c->group, GQ_PREINT, c);
        if (!rc) rc = group_drain(c->group, GQ_WINDOW, c);
        if (rc) return rc;
        group_member_remove(c->group, false);
    }
    c->group = g;
    if (g) group_member_add(g, false);
    return XRHIP_OK;
}

void xrhip_ba_destroy(xrhip_ba *c) {
    if (!c) return;
    if (c->begun.active) xrhip_ba_solve_abort(c);
    if (c->group) {
        xrhip_ba_preintegrate_cancel(c);
        xrhip_ba_join_group(c, nullptr);
    }
    hipStreamSynchronize(c->stream);
    if (c->stream2) hipStreamSynchronize(c->stream2);
    if (std::getenv("XRHIP_HOSTPROF") && c->spec_launched)
        std::fprintf(stderr, "[hostprof] speculative linearisations: %ld launched, %ld taken\n", c->spec_launched, c->spec_taken);
    hipFree(c->work_spec);
    if (c->ev_mid) hipEventDestroy(c->ev_mid);
    if (c->ev_marg) hipEventDestroy(c->ev_marg);
    if (c->ev_spec) hipEventDestroy(c->ev_spec);
    if (c->stream2) hipStreamDestroy(c->stream2);
    hipFree(c->in.dev);
    hipHostFree(c->in.host);
    hipFree(c->work);
    hipFree(c->work2);
    hipHostFree(c->h_stage);
    hipHostFree(c->h_ctl);
    hipHostFree(c->h_seq);
    ba_resolve_pending(c);
    for (auto &e : c->free_events) {
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    hipHostFree(c->h_out);
    hipStreamDestroy(c->stream);
    delete c;
}

static int preint_launch_deferred(xrhip_ba *c, const xrhip_ba_problem *P, const double *state_dev);   // defined with the pre-integration entry points
static int preint_fill_deferred(xrhip_ba *c, const xrhip_ba_problem *P, const double *state_dev, PreintArgs *out, bool *have);

static int ba_solve_impl(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *summary);
int xrhip_ba_solve(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *summary) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve: null context");
    if (c->begun.active) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve: a solve begun on this context has not been collected (xrhip_ba_solve_end)");
    const int rc = ba_solve_impl(c, P, summary);
    if (rc) c->preint_deferred = 0;   // a batch staged behind a solve that failed must not ride on the next, unrelated one
    return rc;
}
int xrhip_ba_solve_overlapped(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *summary, void (*host_work)(void *), void *arg) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_overlapped: null context");
    c->overlap_fn = host_work;
    c->overlap_arg = arg;
    const int rc = xrhip_ba_solve(c, P, summary);
    run_overlap(c);   // no launch was waited for (trivial problem, early error): the work is still the caller's to get done
    return rc;
}
// ---- two solves in one go (xrhip_ba_solve_chained)
static bool any_free_block(const xrhip_ba_problem *P) {
    for (int f = 0; f < P->n_frames; ++f)
        if ((P->frame_fix[f] & 3) != 3) return true;
    return false;
}
static void fill_summary(const BaCtl &ctl, float ms, xrhip_ba_summary *sm) {
    if (!sm) return;
    std::memset(sm, 0, sizeof(*sm));
    sm->iterations = ctl.iteration;
    sm->successful_steps = ctl.successful_steps;
    sm->termination = ctl.termination;
    sm->usable = ctl.termination != XRHIP_BA_FAILURE;
    sm->initial_cost = ctl.initial_cost;
    sm->final_cost = ctl.x_cost;
    sm->ms_solve = ms;
}
// A member of a group does not time its own single-launch solves (the group times the batch they travel in, xrhip_group_stats): with
// profiling on it still accounts their algorithmic bytes -- n_chain_timed / bytes_chain grow, ms_chain does not -- and bench.py
// divides the members' bytes by the group's batch time.
static void chain_account_untimed(xrhip_ba *c, double bytes) {
    if (!c->profiling || !c->group) return;
    c->stats.n_chain_timed++;
    c->stats.bytes_chain += bytes;
}
static double chain_bytes(const BaDims &d, const BaCtl &ctl) {   // algorithmic bytes of a single-launch solve
    const double nf = (double)d.M + d.MR, rounds = ctl.successful_steps + 1.0, trials = ctl.iteration;
    return rounds * (384.0 * nf + 8.0 * (double)d.na * d.na + 2248.0 * d.NI) + trials * (280.0 * nf + 2248.0 * d.NI);
}
static void take_event_pair(xrhip_ba *c, hipEvent_t *e0, hipEvent_t *e1) {
    if (!c->free_events.empty()) {
        *e0 = c->free_events.back().first;
        *e1 = c->free_events.back().second;
        c->free_events.pop_back();
    } else {
        hipEventCreate(e0);
        hipEventCreate(e1);
    }
}

// 1: queued (xrhip_ba_solve_end collects it), 0: not a single-launch solve -- nothing was queued, solve it with xrhip_ba_solve
int xrhip_ba_solve_begin(xrhip_ba *c, const xrhip_ba_problem *P) {
    if (!c || !P) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_begin: null argument");
    if (c->begun.active) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve_begin: a solve is already in flight on this context");
    int rc = validate(P);
    if (rc) return rc;
    static const bool off = std::getenv("XRHIP_NO_CHAINED_SOLVES") != nullptr;   // development switch (A/B, parity)
    // Not for a member of an instance group: there the pair travels as ONE request, and a batch that carries pairs holds the chain
    // queue for two kernels (216 us) while the next batch waits -- two single batches with the host round trip in between serve
    // ten sequences better (5718 -> 5861 frames/s, profiles/r04_multi_sequence.md).  The caller falls back to xrhip_ba_solve.
    static const bool in_group = std::getenv("XRHIP_GROUP_CHAINED_SOLVES") != nullptr;   // development switch (A/B)
    if (off || (c->group && !in_group) || c->preint_deferred || !any_free_block(P)) return 0;
    xrhip_ba::Begun &B = c->begun;
    B.t0 = std::chrono::steady_clock::now();
    Ext cam, imu;
    rc = stage_problem(c, P, B.d, B.p, cam, imu, true);
    if (rc) return rc;
    size_t lds = 0;
    int tile = 0;
    if (!chain(B.d, (size_t)c->lds_limit, &lds, &tile)) return 0;
    if (c->group) {
        rc = group_wait_launched(&c->rq_chain);   // (its argument block is about to be rewritten)
        if (rc) return rc;
    }
    B.seq = ++c->seq;
    B.P = P;
    ChainPayload &cp = c->a_chain;
    cp.stage = StageArgs{c->stage_src, c->stage_dst, c->stage_n16};
    cp.chain = ChainArgs{c->tiny_args, B.seq, 4 * (P->max_iterations + 8), tile, nullptr, -1};
    cp.lds = lds;
    cp.with_preint = false;
    c->rq_chain.kind = GK_CHAIN;
    c->rq_chain.owner = c;
    c->rq_chain.payload = &cp;
    B.e0 = B.e1 = nullptr;
    if (c->group) {
        B.stream = group_stream(c->group, GQ_CHAIN);
        B.rq = &c->rq_chain;
        rc = group_submit(c->group, GQ_CHAIN, &c->rq_chain);
        if (rc) return rc;
    } else {
        B.stream = c->stream;
        B.rq = nullptr;
        if (c->profiling) take_event_pair(c, &B.e0, &B.e1);
        Batch<StageArgs> bs;
        Batch<ChainArgs> bc;
        std::memset(&bs, 0, sizeof(bs));
        std::memset(&bc, 0, sizeof(bc));
        bs.e[0] = cp.stage;
        bc.e[0] = cp.chain;
        hipLaunchKernelGGL(kb_stage, dim3((int)std::min<size_t>((cp.stage.n16 + 255) / 256, 128), 1, 1), dim3(256), 0, B.stream, bs);
        if (B.e0) XR_HIP(hipEventRecord(B.e0, B.stream));
        hipLaunchKernelGGL(kb_chain, dim3(1, 1, 1), dim3(CHAIN_THREADS), lds, B.stream, bc);
        XR_HIP(hipGetLastError());
        if (B.e1) XR_HIP(hipEventRecord(B.e1, B.stream));
    }
    B.active = true;
    B.handed_over = false;
    return 1;
}

int xrhip_ba_solve_end(xrhip_ba *c, xrhip_ba_summary *summary) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_end: null context");
    xrhip_ba::Begun &B = c->begun;
    if (!B.active) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve_end: nothing in flight");
    B.active = false;
    int rc = wait_flag(c->h_seq, B.seq, B.stream, B.rq, "xrhip_ba_solve_end");
    if (rc) return rc;
    if (c->h_ctl->status != ST_DONE) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve_end: trust-region loop did not terminate");
    std::memcpy(B.P->frame_state, c->h_out, sizeof(double) * 16 * B.d.F);
    if (B.e0) c->pending_chain.push_back({B.e0, B.e1, chain_bytes(B.d, *c->h_ctl)});
    else chain_account_untimed(c, chain_bytes(B.d, *c->h_ctl));
    c->stats.n_tiny++;
    kprof_accumulate(*c->h_ctl, B.d.na);
    fill_summary(*c->h_ctl, std::chrono::duration<float, std::milli>((B.handed_over ? B.t_hand : std::chrono::steady_clock::now()) - B.t0).count(),
                 summary);
    B.handed_over = false;
    c->dims = B.d;
    c->ptrs = B.p;
    c->have_lin = true;
    return XRHIP_OK;
}

int xrhip_ba_solve_abort(xrhip_ba *c) {   // the unwind path: wait for a begun solve and forget it (its problem may be gone)
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_abort: null context");
    xrhip_ba::Begun &B = c->begun;
    if (!B.active) return XRHIP_OK;
    B.active = false;
    if (B.rq) group_wait_launched(B.rq);
    XR_HIP(hipStreamSynchronize(B.stream));
    if (B.e0) c->free_events.push_back({B.e0, B.e1});
    return XRHIP_OK;
}

int xrhip_ba_solve_linked(xrhip_ba *c2, const xrhip_ba_problem *P2, xrhip_ba_summary *s2, int link_second, xrhip_ba *c1, int link_first,
                          void (*host_work)(void *), void *arg) {
    if (!c1 || !c2 || c1 == c2 || !P2) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_linked: two distinct contexts and a problem are needed");
    xrhip_ba::Begun &A = c1->begun;
    if (!A.active) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve_linked: the first context has no solve in flight (xrhip_ba_solve_begin)");
    int rc = validate(P2);
    if (rc) return rc;
    if (link_first < 0 || link_first >= A.d.F || link_second < 0 || link_second >= P2->n_frames)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_linked: linked frame is not a frame of its problem");
    // the general form: wait for the first solve, hand the state over on the host, solve the second like any other problem
    auto sequential = [&]() {
        int r = wait_flag(c1->h_seq, A.seq, A.stream, A.rq, "xrhip_ba_solve_linked");
        if (r) return r;
        std::memcpy(P2->frame_state + 16 * (size_t)link_second, c1->h_out + 16 * (size_t)link_first, sizeof(double) * 16);
        return xrhip_ba_solve_overlapped(c2, P2, s2, host_work, arg);
    };
    if (c1->group != c2->group || c2->begun.active || !any_free_block(P2)) return sequential();
    const auto t_begin = std::chrono::steady_clock::now();   // the first solve's clock stops here ONCE the linked request is out (below)
    BaDims d2;
    BaPtrs p2;
    Ext cam2, imu2;
    rc = stage_problem(c2, P2, d2, p2, cam2, imu2, true);
    if (rc) return rc;
    size_t lds2 = 0;
    int tile2 = 0;
    if (!chain(d2, (size_t)c2->lds_limit, &lds2, &tile2)) return sequential();
    if (c2->preint_deferred) {   // refuse a bad frame index before anything is queued (as xrhip_ba_solve does)
        const PreintJob *jobs = (const PreintJob *)(c2->h_stage + c2->preint_o_jobs);
        for (int k = 0; k < c2->preint_deferred; ++k)
            if (jobs[k].bias_frame < 0 || jobs[k].bias_frame >= P2->n_frames) {
                c2->preint_deferred = 0;
                return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_after_solve: bias frame is not a frame of the solve");
            }
    }
    if (c2->group) {
        rc = group_wait_launched(&c2->rq_chain);   // (its argument block is about to be rewritten)
        if (rc) return rc;
    }
    const int seq2 = ++c2->seq;
    ChainPayload &cp = c2->a_chain;
    cp.stage = StageArgs{c2->stage_src, c2->stage_dst, c2->stage_n16};
    cp.chain = ChainArgs{c2->tiny_args, seq2, 4 * (P2->max_iterations + 8), tile2, static_cast<const double *>(A.p.state) + 16 * (size_t)link_first,
                         link_second};
    cp.lds = lds2;
    cp.with_preint = false;
    rc = preint_fill_deferred(c2, P2, p2.state, &cp.preint, &cp.with_preint);
    if (rc) return rc;
    c2->rq_chain.kind = GK_CHAIN;
    c2->rq_chain.owner = c2;
    c2->rq_chain.payload = &cp;
    hipStream_t ps = A.stream;   // behind the first solve, on ITS stream (grouped: the queue both were submitted to, in this order)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c2->group) {
        rc = group_submit(c2->group, GQ_CHAIN, &c2->rq_chain);
        if (rc) return rc;
    } else {
        if (c1->profiling) take_event_pair(c1, &e0, &e1);
        GroupRequest *one = &c2->rq_chain;
        Batch<StageArgs> bs;
        Batch<ChainArgs> bc;
        std::memset(&bs, 0, sizeof(bs));
        std::memset(&bc, 0, sizeof(bc));
        bs.e[0] = cp.stage;
        bc.e[0] = cp.chain;
        hipLaunchKernelGGL(kb_stage, dim3((int)std::min<size_t>((cp.stage.n16 + 255) / 256, 128), 1, 1), dim3(256), 0, ps, bs);
        if (e0) XR_HIP(hipEventRecord(e0, ps));
        hipLaunchKernelGGL(kb_chain, dim3(1, 1, 1), dim3(CHAIN_THREADS), lds2, ps, bc);
        XR_HIP(hipGetLastError());
        if (e1) XR_HIP(hipEventRecord(e1, ps));
        (void)one;
        if (cp.with_preint) {
            GroupRequest pr;
            pr.payload = &cp.preint;
            GroupRequest *ppr = &pr;
            rc = launch_preint_batch(&ppr, 1, ps);
            if (rc) return rc;
        }
    }
    A.handed_over = true;   // submitted: the first solve's summary time runs to t_begin, this solve's clock covers the rest
    A.t_hand = t_begin;
    if (cp.with_preint) {   // the batch queued behind this solve is in flight from here on
        c2->preint_pending = c2->preint_deferred;
        c2->preint_deferred = 0;
        c2->preint_rq = c2->group ? &c2->rq_chain : nullptr;
        c2->preint_stream = ps;
    }
    if (host_work) host_work(arg);   // the caller's work that neither solve depends on, beside both of them
    rc = wait_flag(c2->h_seq, seq2, ps, c2->group ? &c2->rq_chain : nullptr, "xrhip_ba_solve_linked");
    if (rc) return rc;
    if (c2->h_ctl->status != ST_DONE) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve_linked: trust-region loop did not terminate");
    std::memcpy(P2->frame_state, c2->h_out, sizeof(double) * 16 * d2.F);
    if (e0) c1->pending_chain.push_back({e0, e1, chain_bytes(d2, *c2->h_ctl)});
    else chain_account_untimed(c2, chain_bytes(d2, *c2->h_ctl));
    c2->stats.n_tiny++;
    kprof_accumulate(*c2->h_ctl, d2.na);
    fill_summary(*c2->h_ctl, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), s2);
    c2->dims = d2;
    c2->ptrs = p2;
    c2->have_lin = true;
    return XRHIP_OK;
}

int xrhip_ba_solve_chained(xrhip_ba *c1, const xrhip_ba_problem *P1, xrhip_ba_summary *s1, int link_first, xrhip_ba *c2,
                           const xrhip_ba_problem *P2, xrhip_ba_summary *s2, int link_second, void (*host_work)(void *), void *arg) {
    if (!c1 || !c2 || c1 == c2 || !P1 || !P2) return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_chained: two distinct contexts and two problems are needed");
    int rc = validate(P1);
    if (!rc) rc = validate(P2);
    if (rc) return rc;
    if (link_first < 0 || link_first >= P1->n_frames || link_second < 0 || link_second >= P2->n_frames)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_solve_chained: linked frame is not a frame of its problem");
    const int begun = xrhip_ba_solve_begin(c1, P1);
    if (begun < 0) return begun;
    if (begun == 0) {   // not a single-launch solve: one after the other, the state handed over on the host
        rc = xrhip_ba_solve_overlapped(c1, P1, s1, host_work, arg);
        if (rc) return rc;
        std::memcpy(P2->frame_state + 16 * (size_t)link_second, P1->frame_state + 16 * (size_t)link_first, sizeof(double) * 16);
        return xrhip_ba_solve(c2, P2, s2);
    }
    rc = xrhip_ba_solve_linked(c2, P2, s2, link_second, c1, link_first, host_work, arg);
    const int rc1 = xrhip_ba_solve_end(c1, s1);   // (collected in any case: the context must not stay "in flight")
    return rc ? rc : rc1;
}

static int ba_solve_impl(xrhip_ba *c, const xrhip_ba_problem *P, xrhip_ba_summary *summary) {
    int rc = validate(P);
    if (rc) return rc;
    if (c->preint_deferred) {   // a batch staged by xrhip_ba_preintegrate_after_solve: refuse a bad frame index before anything is queued
        const PreintJob *jobs = (const PreintJob *)(c->h_stage + c->preint_o_jobs);
        for (int k = 0; k < c->preint_deferred; ++k)
            if (jobs[k].bias_frame < 0 || jobs[k].bias_frame >= P->n_frames) {
                c->preint_deferred = 0;
                return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_after_solve: bias frame is not a frame of the solve");
            }
    }
    xrhip_ba_summary sm;
    std::memset(&sm, 0, sizeof(sm));
    // trivial problem: nothing to optimise
    bool any_free = false;
    for (int f = 0; f < P->n_frames; ++f)
        if ((P->frame_fix[f] & 3) != 3) any_free = true;
    if (!any_free) {
        std::vector<char> used(std::max(P->n_landmarks, 1), 0);
        for (int o = 0; o < P->n_obs; ++o) used[P->obs_lm[o]] = 1;
        for (int l = 0; l < P->n_landmarks; ++l)
            if (used[l] && !(P->landmark_fix && P->landmark_fix[l])) any_free = true;
    }
    if (!any_free) {
        sm.termination = XRHIP_BA_CONVERGENCE;
        sm.usable = 1;
        if (summary) *summary = sm;
        return preint_launch_deferred(c, P, nullptr);
    }
    BaDims d;
    BaPtrs p;
    Ext cam, imu;
    const auto t_begin = std::chrono::steady_clock::now();
    {
        HostProfScope hp(8, "ba_solve: stage_problem");
        rc = stage_problem(c, P, d, p, cam, imu, true);
    }
    if (rc) return rc;
    size_t chain_lds = 0;
    int chain_tile = 0;
    const bool use_chain = chain(d, (size_t)c->lds_limit, &chain_lds, &chain_tile);
    // Round 5: a member of an instance group hands the rounds of a window-sized solve to the group (GK_WROUND / GK_WTRIALS: one launch per
    // kernel for all members whose rounds are pending, blockIdx.z = member; launch_wround_batch) instead of issuing ~45 launches of its
    // own on a stream that shares two hardware queues with the other members' rounds.  Same kernels' bodies, same bits.
    static const bool wbatch_off = std::getenv("XRHIP_GROUP_NO_WINDOW_BATCH") != nullptr;   // development switch (A/B)
    const bool wbatch = c->group && !use_chain && !wbatch_off && wide_first(d) && d.nla > 0 && std::getenv("XRHIP_GROUP_SPEC") == nullptr;
    if (!use_chain && !wbatch) {   // (the single-launch solve and the group's first round carry the copy of the staged problem themselves)
        rc = launch_stage_copy(c);
        if (rc) return rc;
    }
    HostProfScope hp_rounds(9, "ba_solve: rounds (launch+wait)");
    if (!d.nla) d.lm_rows = 0;
    const double sx = P->sqrt_inv_cov[0], sy = P->sqrt_inv_cov[1];
    hipStream_t s = c->stream;
    if (d.np && !wbatch) hipLaunchKernelGGL(kb_prior_lambda, dim3(((d.np + 15) / 16) * ((d.np + 15) / 16 + 1)), dim3(256), 0, s, d.np, p.pS, p.pLam, p.pinfo, p.pc0);
    bool done = false, relinearise = true;
    bool wfirst = true;   // (wbatch) the next round request is the solve's first
    hipStream_t wstream = wbatch ? group_stream(c->group, GQ_WINDOW) : nullptr;
    const size_t wide_lds = sizeof(double) * ((size_t)WIDE_B * (16 * (size_t)d.F + (size_t)d.np) + (size_t)4 * WIDE_B * 257);
    auto submit_window = [&](int kind, int seq_, int mode_) -> int {
        int r = group_wait_launched(&c->rq_window);   // (its argument block is about to be rewritten)
        if (r) return r;
        WindowPayload &wp = c->a_window;
        size_t lds = 0;
        int use_lds = 1;
        r = solve_lds(d, (size_t)c->lds_limit, &lds, &use_lds);
        if (r) return r;
        lds = std::max(lds, sizeof(double) * (size_t)std::max(TRY_B * (d.np + 15 * d.NI), 1));
        const int tiles = d.PF / 16, nrest = sred_rest_blocks(d);
        wp.e = WinEntry{c->tiny_args, lin_all_blocks(d.M, d.MR, d.NI, d.np), d.lm_rows + d.F * d.F * VIS_CH, (d.n * d.n + 255) / 256,
                        nrest + tiles * tiles + aux_quad_blocks_n(d.n, d.L) + aux_wog_blocks(d.F) + 1, use_lds, mode_, seq_, 1};
        wp.relin = relinearise;
        wp.with_stage = wfirst;
        wp.stage = StageArgs{c->stage_src, c->stage_dst, c->stage_n16};
        wp.np = d.np;
        wp.pS = p.pS;
        wp.pLam = p.pLam;
        wp.pinfo = p.pinfo;
        wp.pc0 = p.pc0;
        wp.lds_lin = sizeof(double) * (size_t)std::max(d.np, 1);
        wp.lds_solve = lds;
        wp.lds_wide = wide_lds;
        c->rq_window.kind = kind;
        c->rq_window.owner = c;
        c->rq_window.payload = &wp;
        wfirst = false;
        return group_submit(c->group, GQ_WINDOW, &c->rq_window);
    };
    int mode = 1, iter_seen = 0;
    struct BusyElsewhere {   // a solve on the context's own stream: the group's linger does not wait for this member meanwhile
        xrhip_group *g;
        explicit BusyElsewhere(xrhip_group *gg) : g(gg) { group_busy_elsewhere(g, 1); }
        ~BusyElsewhere() { group_busy_elsewhere(g, -1); }
    } busy_elsewhere(use_chain ? nullptr : c->group);
    if (use_chain) {   // the whole solve in one launch, LDS-resident (kb_chain)
        const int seq = ++c->seq;
        if (c->group) {
            rc = group_wait_launched(&c->rq_chain);   // (its argument block is about to be rewritten)
            if (rc) return rc;
        }
        ChainPayload &cp = c->a_chain;
        cp.stage = StageArgs{c->stage_src, c->stage_dst, c->stage_n16};
        cp.chain = ChainArgs{c->tiny_args, seq, 4 * (P->max_iterations + 8), chain_tile, nullptr, -1};
        cp.lds = chain_lds;
        // a pre-integration that starts from this solve's biases runs right behind it, reading them where the kernel leaves them
        cp.with_preint = false;
        rc = preint_fill_deferred(c, P, p.state, &cp.preint, &cp.with_preint);
        if (rc) return rc;
        c->rq_chain.kind = GK_CHAIN;
        c->rq_chain.owner = c;
        c->rq_chain.payload = &cp;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipStream_t ps = s;
        if (c->group) {
            ps = group_stream(c->group, GQ_CHAIN);
            rc = group_submit(c->group, GQ_CHAIN, &c->rq_chain);
            if (rc) return rc;
        } else {
            if (c->profiling) {
                if (!c->free_events.empty()) {
                    e0 = c->free_events.back().first;
                    e1 = c->free_events.back().second;
                    c->free_events.pop_back();
                } else {
                    XR_HIP(hipEventCreate(&e0));
                    XR_HIP(hipEventCreate(&e1));
                }
            }
            // launched here: kb_stage, [event] kb_chain [event], the queued integration -- the events bracket kb_chain alone
            Batch<StageArgs> bs;
            Batch<ChainArgs> bc;
            std::memset(&bs, 0, sizeof(bs));
            std::memset(&bc, 0, sizeof(bc));
            bs.e[0] = cp.stage;
            bc.e[0] = cp.chain;
            hipLaunchKernelGGL(kb_stage, dim3((int)std::min<size_t>((cp.stage.n16 + 255) / 256, 128), 1, 1), dim3(256), 0, s, bs);
            if (e0) XR_HIP(hipEventRecord(e0, s));
            hipLaunchKernelGGL(kb_chain, dim3(1, 1, 1), dim3(CHAIN_THREADS), chain_lds, s, bc);
            XR_HIP(hipGetLastError());
            if (e1) XR_HIP(hipEventRecord(e1, s));
            if (cp.with_preint) {
                GroupRequest one;
                one.payload = &cp.preint;
                GroupRequest *pone = &one;
                rc = launch_preint_batch(&pone, 1, s);
                if (rc) return rc;
            }
        }
        if (cp.with_preint) {   // the batch is in flight from here on (xrhip_ba_preintegrate_end collects it)
            c->preint_pending = c->preint_deferred;
            c->preint_deferred = 0;
            c->preint_rq = c->group ? &c->rq_chain : nullptr;
            c->preint_stream = ps;
        }
        rc = wait_mailbox(c, seq, ps, c->group ? &c->rq_chain : nullptr);
        if (rc) return rc;
        if (c->h_ctl->status != ST_DONE) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve: trust-region loop did not terminate");
        if (e0) {   // algorithmic bytes of this launch: rounds = accepted steps + 1 linearisations, `iteration` candidates
            const double nf = (double)d.M + d.MR, rounds = c->h_ctl->successful_steps + 1.0, trials = c->h_ctl->iteration;
            const double bytes = rounds * (384.0 * nf + 8.0 * (double)d.na * d.na + 2248.0 * d.NI) + trials * (280.0 * nf + 2248.0 * d.NI);
            c->pending_chain.push_back({e0, e1, bytes});
        } else {
            chain_account_untimed(c, chain_bytes(d, *c->h_ctl));
        }
        c->stats.n_tiny++;
        done = true;
    } else if (tiny(d)) {   // the whole trust-region loop in one launch (kb_tiny)
        const int seq = ++c->seq;
        size_t lds = 0;
        int use_lds = 1;
        rc = solve_lds(d, (size_t)c->lds_limit, &lds, &use_lds);
        if (rc) return rc;
        lds = std::max(lds, sizeof(double) * (size_t)std::max(TRY_B * (d.np + 15 * d.NI), 1));
        hipLaunchKernelGGL(kb_tiny, dim3(1), dim3(256), lds, s, c->tiny_args, use_lds, seq, 4 * (P->max_iterations + 8));
        XR_HIP(hipGetLastError());
        rc = wait_mailbox(c, seq);
        if (rc) return rc;
        if (c->h_ctl->status != ST_DONE) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve: trust-region loop did not terminate");
        c->stats.n_tiny++;
        done = true;
    }
    // ---- speculative linearisation (window solves, see spec_state_block in ba_kernels.hip.h): the second buffer set mirrors the workspace
    static const bool spec_off = std::getenv("XRHIP_NO_SPEC") != nullptr;   // development switch
    static const bool spec_in_group = std::getenv("XRHIP_GROUP_SPEC") != nullptr;   // development switch (A/B)
    // Not in an instance group: the speculation buys one sequence ~0.1 ms per keyframe with a second stream and four extra launches per
    // round -- with several sequences on the device those launches take queue turns from the other members' kernels (8 sequences:
    // 3996 -> 4326 frames/s without it, profiles/r04_multi_sequence.md).  (Same results either way: the speculated linearisation is
    // the one the round would have computed.)
    const bool spec = !spec_off && !use_chain && wide_first(d) && !(c->group && !spec_in_group);
    BaPtrs p2 = p;
    if (spec) {
        const size_t extra = sizeof(double) * (16 * (size_t)d.F + (size_t)std::max(d.L, 1)) + sizeof(BaCtl) + 1024;
        if (c->work_spec_cap < c->work_cap + extra) {
            if (c->work_spec) hipFree(c->work_spec);
            c->work_spec = nullptr;
            XR_HIP(hipMalloc(&c->work_spec, c->work_cap + extra));
            c->work_spec_cap = c->work_cap + extra;
        }
        auto shift = [&](auto &member) {
            using T = std::remove_reference_t<decltype(*(static_cast<decltype(&*member)>(nullptr)))>;
            member = reinterpret_cast<T *>(c->work_spec + (reinterpret_cast<const char *>(static_cast<T *>(member)) - c->work));
        };
        shift(p2.orec); shift(p2.ocost); shift(p2.rrec); shift(p2.rcost);
        shift(p2.imu_r); shift(p2.imu_Ji); shift(p2.imu_Jj); shift(p2.imu_cost);
        shift(p2.pr); shift(p2.pt); shift(p2.pJq); shift(p2.pcost);
        shift(p2.hll); shift(p2.gl); shift(p2.Wt); shift(p2.Hv); shift(p2.gv);
        shift(p2.Hpp); shift(p2.gp); shift(p2.diagD); shift(p2.gs); shift(p2.omega);
        shift(p2.T); shift(p2.Sred); shift(p2.partial); shift(p2.wog);
        char *x = c->work_spec + ((c->work_cap + 255) & ~size_t(255));
        p2.state = reinterpret_cast<double *>(x);
        p2.depth = reinterpret_cast<double *>(x + sizeof(double) * 16 * (size_t)d.F);
        p2.ctl = reinterpret_cast<BaCtl *>(x + ((sizeof(double) * (16 * (size_t)d.F + (size_t)std::max(d.L, 1)) + 255) & ~size_t(255)));
    }
    auto swap_sets = [&]() {   // the products of the linearisation chain change places; state / depth / ctl stay the minimiser's
        std::swap(p.orec, p2.orec); std::swap(p.ocost, p2.ocost); std::swap(p.rrec, p2.rrec); std::swap(p.rcost, p2.rcost);
        std::swap(p.imu_r, p2.imu_r); std::swap(p.imu_Ji, p2.imu_Ji); std::swap(p.imu_Jj, p2.imu_Jj); std::swap(p.imu_cost, p2.imu_cost);
        std::swap(p.pr, p2.pr); std::swap(p.pt, p2.pt); std::swap(p.pJq, p2.pJq); std::swap(p.pcost, p2.pcost);
        std::swap(p.hll, p2.hll); std::swap(p.gl, p2.gl); std::swap(p.Wt, p2.Wt); std::swap(p.Hv, p2.Hv); std::swap(p.gv, p2.gv);
        std::swap(p.Hpp, p2.Hpp); std::swap(p.gp, p2.gp); std::swap(p.diagD, p2.diagD); std::swap(p.gs, p2.gs); std::swap(p.omega, p2.omega);
        std::swap(p.T, p2.T); std::swap(p.Sred, p2.Sred); std::swap(p.partial, p2.partial); std::swap(p.wog, p2.wog);
    };
    bool spec_ready = false;   // the linearisation at the state just accepted is already queued on the second stream
    for (int guard = 0; guard < 4 * (P->max_iterations + 8) && !done; ++guard) {
        const int seq = ++c->seq;
        if (wbatch) {
            rc = submit_window(GK_WROUND, seq, mode);   // (timed by the group as part of its GK_WROUND batch: no event pair, no count here)
        } else if (spec_ready) {   // (the chain that built this linearisation is ahead of us on this very stream)
            rc = launch_solve_try(c, d, p, cam, imu, sx, sy, false, mode, seq, true, &p2);
        } else {
            if (relinearise) launch_linearize(c, d, p, cam, imu, sx, sy, true);
            rc = launch_solve_try(c, d, p, cam, imu, sx, sy, !relinearise, mode, seq, false, spec ? &p2 : nullptr);
        }
        if (rc) return rc;
        spec_ready = false;
        bool spec_launched = false;
        if (spec) {   // behind the factorisation, beside kb_trials_wide (second stream): the problem linearised at the first candidate
            BaPtrs q = p2;             // the chain reads the candidate as "the state"
            launch_linearize(c, d, q, cam, imu, sx, sy, true);
            launch_schur_aux(d, q, s);
            XR_HIP(hipGetLastError());
            spec_launched = true;
            c->spec_launched++;
        }
        rc = wbatch ? wait_mailbox(c, seq, wstream, &c->rq_window)
                    : wait_mailbox(c, seq, spec ? c->stream2 : nullptr);   // no copy, no driver wait: the kernel's last store is the sequence number
        if (rc) return rc;
        const bool first_batch_accept = spec_launched && c->h_ctl->status == ST_ACCEPTED && c->h_ctl->accepted_slot == 0;
        {   // trials this launch costed = trust-region iterations it advanced
            const int it_now = c->h_ctl->iteration;
            const int trials = std::max(0, it_now - iter_seen);
            iter_seen = it_now;
            c->stats.n_trials += trials;
            if (c->profiling && !wbatch && !c->pending.empty()) c->pending.back().iter_before = trials;   // this launch's own event pair
        }
        int st = c->h_ctl->status;
        for (int wguard = 0; st == ST_NEED_TRIALS && wguard < 64; ++wguard) {   // run of rejected trials, 8 per launch
            const int wseq = ++c->seq;
            if (wbatch) {
                rc = submit_window(GK_WTRIALS, wseq, 0);
                if (rc) return rc;
                rc = wait_mailbox(c, wseq, wstream, &c->rq_window);
            } else {
                hipLaunchKernelGGL(kb_trials_wide, dim3(WIDE_G), dim3(256), wide_lds, spec ? c->stream2 : s, d, p, cam, imu, sx, sy, wseq, 0, 0);
                XR_HIP(hipGetLastError());
                rc = wait_mailbox(c, wseq, spec ? c->stream2 : nullptr);
            }
            if (rc) return rc;
            const int it_now = c->h_ctl->iteration;
            c->stats.n_trials += std::max(0, it_now - iter_seen);
            iter_seen = it_now;
            st = c->h_ctl->status;
        }
        if (st == ST_DONE) {
            done = true;   // the optimised states are already in h_out
        } else if (st == ST_ACCEPTED) {
            relinearise = true;
            mode = 1;
            if (first_batch_accept) {   // the speculation linearised exactly this state: take its buffers
                swap_sets();
                spec_ready = true;
                c->spec_taken++;
            }
        } else if (st == ST_RESOLVE || st == ST_RESOLVE_INNER) {
            relinearise = false;
            mode = (st == ST_RESOLVE) ? 2 : 3;
        } else {
            return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve: unexpected device status");
        }
    }
    if (!done) return xr_fail(XRHIP_ESTATE, "xrhip_ba_solve: trust-region loop did not terminate");
    // the optimised states (in place, like the reference)
    std::memcpy(P->frame_state, c->h_out, sizeof(double) * 16 * d.F);
    if (d.L && d.nla) std::memcpy(P->inv_depth, c->h_out + 16 * d.F, sizeof(double) * d.L);   // no free landmark: nothing moved
    rc = preint_launch_deferred(c, P, nullptr);   // (the single-launch path has launched it already)
    if (rc) return rc;
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    const BaCtl &ctl = *c->h_ctl;
    kprof_accumulate(ctl, d.na);
    sm.iterations = ctl.iteration;
    sm.successful_steps = ctl.successful_steps;
    sm.termination = ctl.termination;
    sm.usable = ctl.termination != XRHIP_BA_FAILURE;
    sm.initial_cost = ctl.initial_cost;
    sm.final_cost = ctl.x_cost;
    sm.ms_solve = ms;
    if (summary) *summary = sm;
    c->dims = d;
    c->ptrs = p;
    c->have_lin = true;
    return XRHIP_OK;
}

/* study aid (BASELINE config 5, "fp32 vs bf16 BA solve"): the Schur contraction of every following solve on this context runs with f32
 * (mode 1) or bf16 (mode 2) matrix-core operands instead of f64 (mode 0, what the product always uses). */
int xrhip_ba_debug_marg_guard(xrhip_ba *c, double *lambda_bound, int *status8) {
    if (!c || !lambda_bound || !status8) return xr_fail(XRHIP_EINVAL, "xrhip_ba_debug_marg_guard: null argument");
    if (c->marg.pending || !c->marg.lam) return xr_fail(XRHIP_ESTATE, "xrhip_ba_debug_marg_guard: no collected marginalisation on this context");
    XR_HIP(hipStreamSynchronize(c->stream));
    XR_HIP(hipMemcpy(lambda_bound, c->marg.lam, sizeof(double), hipMemcpyDeviceToHost));
    XR_HIP(hipMemcpy(status8, c->marg.dst, sizeof(int) * 8, hipMemcpyDeviceToHost));
    return XRHIP_OK;
}

int xrhip_ba_debug_set_schur_precision(xrhip_ba *c, int mode) {
    if (!c || mode < 0 || mode > 2) return xr_fail(XRHIP_EINVAL, "xrhip_ba_debug_set_schur_precision: bad arguments");
    c->schur_mode = mode;
    return XRHIP_OK;
}

/* parity/testing aid: linearise the problem at its current states (no solve) and return the unreduced
 * normal equations in "frame-major" layout: H [15F x 15F], g [15F], hll [L], gl [L], W [L x 6F] (row l =
 * cross terms of landmark l with the pose dofs of every frame), cost.  Constant blocks have zero rows. */
int xrhip_ba_debug_linearize(xrhip_ba *c, const xrhip_ba_problem *P, double *H, double *g, double *hll, double *gl,
                             double *W, double *cost) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_debug_linearize: null context");
    int rc = validate(P);
    if (rc) return rc;
    BaDims d;
    BaPtrs p;
    Ext cam, imu;
    rc = stage_problem(c, P, d, p, cam, imu);
    if (rc) return rc;
    hipStream_t s = c->stream;
    if (d.np) hipLaunchKernelGGL(kb_prior_lambda, dim3(((d.np + 15) / 16) * ((d.np + 15) / 16 + 1)), dim3(256), 0, s, d.np, p.pS, p.pLam, p.pinfo, p.pc0);
    launch_linearize(c, d, p, cam, imu, P->sqrt_inv_cov[0], P->sqrt_inv_cov[1], false);
    XR_HIP(hipGetLastError());
    XR_HIP(hipMemcpyAsync(c->h_ctl, p.ctl, sizeof(BaCtl), hipMemcpyDeviceToHost, s));
    if (H) XR_HIP(hipMemcpyAsync(H, p.Hpp, sizeof(double) * (size_t)d.n * d.n, hipMemcpyDeviceToHost, s));
    if (g) XR_HIP(hipMemcpyAsync(g, p.gp, sizeof(double) * d.n, hipMemcpyDeviceToHost, s));
    if (hll && d.L) XR_HIP(hipMemcpyAsync(hll, p.hll, sizeof(double) * d.L, hipMemcpyDeviceToHost, s));
    if (gl && d.L) XR_HIP(hipMemcpyAsync(gl, p.gl, sizeof(double) * d.L, hipMemcpyDeviceToHost, s));
    std::vector<double> wt;
    if (W && d.L) {
        wt.resize((size_t)d.Lp * d.PF);
        XR_HIP(hipMemcpyAsync(wt.data(), p.Wt, sizeof(double) * wt.size(), hipMemcpyDeviceToHost, s));
    }
    XR_HIP(hipStreamSynchronize(s));
    if (W && d.L)
        for (int l = 0; l < d.L; ++l)
            for (int a = 0; a < 6 * d.F; ++a) W[(size_t)l * 6 * d.F + a] = wt[(size_t)l * d.PF + a];
    if (cost) *cost = c->h_ctl->x_cost;
    return XRHIP_OK;
}

/* parity/testing aid: T = W^T diag(w) W through the MFMA kernel for arbitrary inputs.
 * W: [L][P] row-major, w: [L]; out: [P][P].  P is padded to 16 and L to 16 internally. */
int xrhip_ba_debug_schur(xrhip_ba *c, const double *W, const double *w, int L, int P, double *out) {
    if (!c || !W || !w || !out || L <= 0 || P <= 0) return xr_fail(XRHIP_EINVAL, "xrhip_ba_debug_schur: bad arguments");
    BaDims d;
    std::memset(&d, 0, sizeof(d));
    BaPtrs p;
    std::memset(&p, 0, sizeof(p));
    d.PF = round_up(P, 16);
    d.L = L;
    d.Lp = std::max(16, round_up(L, 16));
    std::vector<double> wt((size_t)d.Lp * d.PF, 0.0), om(d.Lp, 0.0);
    for (int l = 0; l < L; ++l) {
        om[l] = w[l];
        for (int a = 0; a < P; ++a) wt[(size_t)l * d.PF + a] = W[(size_t)l * P + a];
    }
    double *dW = nullptr, *dw = nullptr, *dT = nullptr;
    XR_HIP(hipMalloc(&dW, sizeof(double) * wt.size()));
    XR_HIP(hipMalloc(&dw, sizeof(double) * om.size()));
    XR_HIP(hipMalloc(&dT, sizeof(double) * (size_t)d.PF * d.PF));
    XR_HIP(hipMemcpy(dW, wt.data(), sizeof(double) * wt.size(), hipMemcpyHostToDevice));
    XR_HIP(hipMemcpy(dw, om.data(), sizeof(double) * om.size(), hipMemcpyHostToDevice));
    p.Wt = dW;
    p.omega = dw;
    p.T = dT;
    const int tiles = d.PF / 16;
    hipLaunchKernelGGL(kb_schur_mfma, dim3(tiles * tiles), dim3(256), 0, c->stream, d, p);
    XR_HIP(hipGetLastError());
    XR_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> T((size_t)d.PF * d.PF);
    XR_HIP(hipMemcpy(T.data(), dT, sizeof(double) * T.size(), hipMemcpyDeviceToHost));
    for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) out[(size_t)a * P + b] = T[(size_t)a * d.PF + b];
    hipFree(dW);
    hipFree(dw);
    hipFree(dT);
    return XRHIP_OK;
}

}   // extern "C"

static int ensure_work2(xrhip_ba *c, size_t dev_bytes, size_t host_bytes) {
    if (dev_bytes > c->work2_cap) {
        if (c->work2) hipFree(c->work2);
        c->work2 = nullptr;
        size_t cap = std::max(dev_bytes * 2, size_t(1) << 20);
        XR_HIP(hipMalloc(&c->work2, cap));
        c->work2_cap = cap;
    }
    if (host_bytes > c->h_stage_cap) {
        if (c->h_stage) hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        size_t cap = std::max(host_bytes * 2, size_t(1) << 20);
        XR_HIP(hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault));
        c->h_stage_cap = cap;
    }
    return XRHIP_OK;
}

extern "C" {

// Marginalisation in two halves.  marg_launch stages the problem and queues EVERY kernel and copy of the Cholesky
// path (the common one) without waiting; marg_collect waits, re-does the tail through the eigen-solver if the device
// reported that the Cholesky path does not apply, and hands the prior out.
// status words of a marginalisation (device, 8 ints): [0] singular victim block, [1] support size, [2] Cholesky failed (later: the
// sweeps of the eigen path), [3] eigenvalue guard failed, [4] the eigen path was taken, [5] the support size km_jacobi sees
// (round 5: the gate -- st[4] = the eigen path is needed, st[5] = its support size -- is the tail of km_chol, marg_kernels.hip.h)

static int marg_launch(xrhip_ba *c, const xrhip_marg_problem *M) {
    if (c->marg.pending) return xr_fail(XRHIP_ESTATE, "xrhip_ba_marginalize: a marginalisation is already in flight");
    if (M->n_frames < 2 || M->victim < 0 || M->victim >= M->n_frames)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_marginalize: bad frame count / victim");
    // express the marginalisation's linearisation as a BA problem with every block free and no robust loss
    const int K = M->n_frames;
    std::vector<double> state(M->frame_state, M->frame_state + 16 * (size_t)K);
    std::vector<uint8_t> fix(K, 0);
    std::vector<double> depth(M->inv_depth, M->inv_depth + std::max(M->n_landmarks, 0));
    std::vector<uint8_t> lfix(std::max(M->n_landmarks, 1), 0);
    xrhip_ba_problem P;
    std::memset(&P, 0, sizeof(P));
    P.n_frames = K;
    P.frame_state = state.data();
    P.frame_fix = fix.data();
    std::memcpy(P.cam_q_bc, M->cam_q_bc, sizeof(P.cam_q_bc));
    std::memcpy(P.cam_p_bc, M->cam_p_bc, sizeof(P.cam_p_bc));
    std::memcpy(P.imu_q_bi, M->imu_q_bi, sizeof(P.imu_q_bi));
    std::memcpy(P.imu_p_bi, M->imu_p_bi, sizeof(P.imu_p_bi));
    std::memcpy(P.sqrt_inv_cov, M->sqrt_inv_cov, sizeof(P.sqrt_inv_cov));
    P.n_landmarks = M->n_landmarks;
    P.inv_depth = depth.data();
    P.landmark_fix = lfix.data();
    P.n_obs = M->n_obs;
    P.obs_tgt = M->obs_tgt;
    P.obs_ref = M->obs_ref;
    P.obs_lm = M->obs_lm;
    P.obs_z_tgt = M->obs_z_tgt;
    P.obs_z_ref = M->obs_z_ref;
    P.n_imu = M->n_imu;
    P.imu_i = M->imu_i;
    P.imu_j = M->imu_j;
    P.imu_data = M->imu_data;
    P.prior_n = M->prior_n;
    P.prior_frames = M->prior_frames;
    P.prior_sqrt_info = M->prior_sqrt_info;
    P.prior_infovec = M->prior_infovec;
    P.prior_lin = M->prior_lin;
    P.max_iterations = 0;
    int rc = validate(&P);
    if (rc) return rc;
    BaDims d;
    BaPtrs p;
    Ext cam, imu;
    rc = stage_problem(c, &P, d, p, cam, imu);
    if (rc) return rc;
    d.robust = 0;
    d.schur_mode = 0;
    const int N = d.n, R = N - 15;
    if (R > 512) return xr_fail(XRHIP_EINVAL, "xrhip_ba_marginalize: window too large");   // before anything is queued
    const size_t D8 = sizeof(double);
    size_t w = 0;
    auto carve = [&](size_t bytes) {
        size_t off = (w + 255) & ~size_t(255);
        w = off + std::max(bytes, size_t(8));
        return off;
    };
    const size_t o_Hm = carve(D8 * (size_t)N * N), o_bm = carve(D8 * N), o_T2 = carve(D8 * (size_t)R * 15);
    const size_t o_A = carve(D8 * (size_t)R * R), o_bp = carve(D8 * R), o_B = carve(D8 * (size_t)R * R);
    const size_t o_V = carve(D8 * (size_t)R * R), o_si = carve(D8 * (size_t)R * R), o_iv = carve(D8 * R);
    const size_t o_st = carve(sizeof(int) * 8), o_lam = carve(D8 * 2), o_sup = carve(sizeof(int) * (size_t)R);
    const size_t o_As = carve(D8 * (size_t)R * R), o_bs = carve(D8 * R), o_Ss = carve(D8 * (size_t)R * R), o_ivs = carve(D8 * R);
    rc = ensure_work2(c, w + 256, D8 * ((size_t)R * R + R) + 64 + sizeof(int) * 8);
    if (rc) return rc;
    char *W2 = c->work2;
    double *Hm = (double *)(W2 + o_Hm), *bm = (double *)(W2 + o_bm), *T2 = (double *)(W2 + o_T2);
    double *A = (double *)(W2 + o_A), *bp = (double *)(W2 + o_bp), *B = (double *)(W2 + o_B), *V = (double *)(W2 + o_V);
    double *dsi = (double *)(W2 + o_si), *div = (double *)(W2 + o_iv);
    int *dst = (int *)(W2 + o_st);
    hipStream_t s = c->stream;
    XR_HIP(hipMemsetAsync(dst, 0, sizeof(int) * 8, s));
    if (d.np) hipLaunchKernelGGL(kb_prior_lambda, dim3(((d.np + 15) / 16) * ((d.np + 15) / 16 + 1)), dim3(256), 0, s, d.np, p.pS, p.pLam, p.pinfo, p.pc0);
    launch_linearize(c, d, p, cam, imu, M->sqrt_inv_cov[0], M->sqrt_inv_cov[1], false);
    hipLaunchKernelGGL(km_omega, dim3((std::max(d.L, 1) + 255) / 256), dim3(256), 0, s, d, p);
    const int tiles = d.PF / 16;
    hipLaunchKernelGGL(kb_schur_mfma, dim3(tiles * tiles), dim3(256), 0, s, d, p);
    hipLaunchKernelGGL(km_wog, dim3(aux_wog_blocks(d.F)), dim3(256), 0, s, d, p);
    hipLaunchKernelGGL(km_permute, dim3((N * N + 255) / 256), dim3(256), 0, s, d, p, M->victim, Hm, bm);
    hipLaunchKernelGGL(km_victim, dim3(1), dim3(256), 0, s, N, Hm, T2, dst);
    hipLaunchKernelGGL(km_complement, dim3((R * R + 255) / 256), dim3(256), 0, s, N, Hm, bm, T2, A, bp);
    double *hs = (double *)c->h_stage;
    int *dsup = (int *)(W2 + o_sup), *dsn = dst + 1;
    double *As = (double *)(W2 + o_As), *bs = (double *)(W2 + o_bs), *Ss = (double *)(W2 + o_Ss), *ivs = (double *)(W2 + o_ivs);
    const int lds_doubles = c->lds_limit / (int)D8;
    hipLaunchKernelGGL(km_support, dim3(1), dim3(512), 0, s, R, A, bp, dsup, dsn, As, bs);
    // fast path: Cholesky factor of the compacted matrix as sqrt_info (valid when no eigenvalue is near the 1e-8 floor)
    hipLaunchKernelGGL(km_chol, dim3(1), dim3(512), (size_t)c->lds_limit, s, dsn, lds_doubles, As, bs, dsup, R, dsi, div,
                       (double *)(W2 + o_lam), dst);
    XR_HIP(hipGetLastError());
    // The eigen path (the reference's SelfAdjointEigenSolver route, km_jacobi: milliseconds) is needed when the Cholesky
    // factor failed or an eigenvalue sits near the 1e-8 floor -- typically the first marginalisation of a sequence.  It used to
    // be started by the HOST when it collected the result, i.e. on the critical path of the next refine_window.  Now the
    // device decides: km_chol's tail hands km_jacobi the support size if the fast path's status words ask for the fallback, and 0
    // otherwise (km_jacobi returns at once on an empty support), so the fallback runs behind the fast path on this context's
    // own stream like the rest of the marginalisation.  Then one expansion of whichever factor is in place, and the copies.
    // round 5: km_chol ends with the gate and has already expanded its own factor; km_jacobi (an empty launch unless the gate asks for
    // it) expands its own -- two launches instead of five behind the factorisation
    hipLaunchKernelGGL(km_jacobi, dim3(1), dim3(1024), (size_t)c->lds_limit, s, dst + 5, lds_doubles, As, bs, B, V, Ss, ivs, 60, dst + 2,
                       dsup, R, dsi, div);
    XR_HIP(hipGetLastError());
    XR_HIP(hipMemcpyAsync(hs, dsi, D8 * (size_t)R * R, hipMemcpyDeviceToHost, s));
    XR_HIP(hipMemcpyAsync(hs + (size_t)R * R, div, D8 * R, hipMemcpyDeviceToHost, s));
    XR_HIP(hipMemcpyAsync(hs + (size_t)R * R + R, dst, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
    if (!c->ev_marg) XR_HIP(hipEventCreateWithFlags(&c->ev_marg, hipEventDisableTiming));
    XR_HIP(hipEventRecord(c->ev_marg, s));
    xrhip_ba::MargPending &mp = c->marg;
    mp.pending = true;
    mp.K = K;
    mp.victim = M->victim;
    mp.R = R;
    mp.lin.assign(M->frame_state, M->frame_state + 16 * (size_t)K);
    mp.dst = dst;
    mp.dsup = dsup;
    mp.lam = (double *)(W2 + o_lam);
    mp.As = As;
    mp.bs = bs;
    mp.B = B;
    mp.V = V;
    mp.Ss = Ss;
    mp.ivs = ivs;
    mp.dsi = dsi;
    mp.div = div;
    return XRHIP_OK;
}

static int marg_collect(xrhip_ba *c, double *out_sqrt_info, double *out_infovec, double *out_lin) {
    xrhip_ba::MargPending &mp = c->marg;
    if (!mp.pending) return xr_fail(XRHIP_ESTATE, "xrhip_ba_marginalize_end: nothing in flight");
    mp.pending = false;
    hipStream_t s = c->stream;
    const size_t D8 = sizeof(double);
    const int R = mp.R, K = mp.K;
    double *hs = (double *)c->h_stage;
    XR_HIP(hipEventSynchronize(c->ev_marg));
    int hst[8];
    std::memcpy(hst, hs + (size_t)R * R + R, sizeof(hst));
    if (hst[0]) return xr_fail(XRHIP_ESTATE, "xrhip_ba_marginalize: singular victim block");
    if (hst[4] && std::getenv("XRHIP_HOSTPROF"))
        std::fprintf(stderr, "[hostprof] marginalisation took the eigen path on the device: guard %s, support %d of %d, %d sweeps\n",
                     hst[3] ? "failed" : "ok", hst[1], R, hst[2]);
    if ((hst[2] != 0 || hst[3] != 0) && !hst[4]) {   // (not reached: the gate kernel has made this decision on the device)
        if (std::getenv("XRHIP_HOSTPROF"))
            std::fprintf(stderr, "[hostprof] marginalisation falls back to the eigen path: cholesky %s, guard %s, support %d of %d\n",
                         hst[2] ? "failed" : "ok", hst[3] ? "failed" : "ok", hst[1], R);
        const int lds_doubles = c->lds_limit / (int)D8;
        int *dsn = mp.dst + 1;
        hipLaunchKernelGGL(km_jacobi, dim3(1), dim3(1024), (size_t)c->lds_limit, s, dsn, lds_doubles, mp.As, mp.bs, mp.B, mp.V, mp.Ss,
                           mp.ivs, 60, mp.dst + 2);
        hipLaunchKernelGGL(km_expand, dim3((R * R + 255) / 256), dim3(256), 0, s, R, mp.dsup, dsn, mp.Ss, mp.ivs, mp.dsi, mp.div);
        XR_HIP(hipGetLastError());
        XR_HIP(hipMemcpyAsync(hs, mp.dsi, D8 * (size_t)R * R, hipMemcpyDeviceToHost, s));
        XR_HIP(hipMemcpyAsync(hs + (size_t)R * R, mp.div, D8 * R, hipMemcpyDeviceToHost, s));
        XR_HIP(hipStreamSynchronize(s));
    }
    std::memcpy(out_sqrt_info, hs, D8 * (size_t)R * R);
    std::memcpy(out_infovec, hs + (size_t)R * R, D8 * R);
    int j = 0;
    for (int i = 0; i < K; ++i) {
        if (i == mp.victim) continue;
        std::memcpy(out_lin + 16 * (size_t)j, mp.lin.data() + 16 * (size_t)i, D8 * 16);
        ++j;
    }
    return XRHIP_OK;
}

int xrhip_ba_marginalize(xrhip_ba *c, const xrhip_marg_problem *M, double *out_sqrt_info, double *out_infovec,
                         double *out_lin) {
    if (!c || !M || !out_sqrt_info || !out_infovec || !out_lin) return xr_fail(XRHIP_EINVAL, "xrhip_ba_marginalize: null argument");
    int rc = marg_launch(c, M);
    if (rc) return rc;
    return marg_collect(c, out_sqrt_info, out_infovec, out_lin);
}

int xrhip_ba_marginalize_begin(xrhip_ba *c, const xrhip_marg_problem *M) {
    if (!c || !M) return xr_fail(XRHIP_EINVAL, "xrhip_ba_marginalize_begin: null argument");
    return marg_launch(c, M);
}

int xrhip_ba_marginalize_end(xrhip_ba *c, double *out_sqrt_info, double *out_infovec, double *out_lin) {
    if (!c || !out_sqrt_info || !out_infovec || !out_lin) return xr_fail(XRHIP_EINVAL, "xrhip_ba_marginalize_end: null argument");
    return marg_collect(c, out_sqrt_info, out_infovec, out_lin);
}

// Stages a batch in the pinned block (jobs, samples, noise; results and status words behind them).  bias_frame: per job, the
// frame of the next solve whose biases the integration starts from (the values in bg / ba are then ignored), or nullptr.
static int preint_stage(xrhip_ba *c, const double *samples, const int *sample_begin, const int *sample_count,
                        const double *t_end, const double *bg, const double *ba, const int *bias_frame, int n_jobs,
                        const double *noise_cov36) {
    // one batch in flight per context: a second begin would overwrite the staging block the first one's kernel reads and
    // writes, and the first owner's _end would collect the wrong record.  An owner that unwinds on an error between begin
    // and end releases the context with xrhip_ba_preintegrate_cancel.
    if (c->preint_pending || c->preint_deferred)
        return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate_begin: a batch is already in flight on this context");
    int total = 0;
    for (int k = 0; k < n_jobs; ++k) {
        if (sample_count[k] <= 0 || sample_begin[k] < 0) return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate: empty IMU segment");
        total = std::max(total, sample_begin[k] + sample_count[k]);
    }
    // zero-copy: jobs, samples, noise, results and status live in the pinned staging block, which the kernel
    // addresses directly over the host link -- an integration moves a few hundred bytes in and 2.2 KB out, so
    // copy-engine round trips (H2D, memset, D2H) would cost several times the kernel itself.  Each job's status word
    // is its completion mailbox.
    const size_t D8 = sizeof(double);
    const size_t b_jobs = sizeof(PreintJob) * n_jobs, b_smp = D8 * 7 * (size_t)total, b_noise = D8 * 36;
    const size_t o_jobs = 0, o_smp = (b_jobs + 255) & ~size_t(255), o_noise = (o_smp + b_smp + 255) & ~size_t(255);
    const size_t o_out = (o_noise + b_noise + 255) & ~size_t(255), o_st = o_out + D8 * XRHIP_IMU_DIM * (size_t)n_jobs;
    const size_t bytes = o_st + 2 * sizeof(int) * n_jobs + 256;   // status words, then the early-delta words
    int rc = ensure_work2(c, 0, bytes);
    if (rc) return rc;
    char *H = c->h_stage;
    PreintJob *jobs = (PreintJob *)(H + o_jobs);
    for (int k = 0; k < n_jobs; ++k) {
        jobs[k].sample_begin = sample_begin[k];
        jobs[k].sample_count = sample_count[k];
        jobs[k].t_end = t_end[k];
        for (int i = 0; i < 3; ++i) {
            jobs[k].bg[i] = bg ? bg[3 * k + i] : 0.0;
            jobs[k].ba[i] = ba ? ba[3 * k + i] : 0.0;
        }
        jobs[k].bias_frame = bias_frame ? bias_frame[k] : -1;
        jobs[k].pad_ = 0;
    }
    std::memcpy(H + o_smp, samples, b_smp);
    std::memcpy(H + o_noise, noise_cov36, b_noise);
    std::memset(H + o_st, 0, 2 * sizeof(int) * n_jobs);
    c->preint_o_jobs = o_jobs;
    c->preint_o_smp = o_smp;
    c->preint_o_noise = o_noise;
    c->preint_o_out = o_out;
    c->preint_o_st = o_st;
    return XRHIP_OK;
}

static int preint_args(xrhip_ba *c, int n_jobs, int jac, int cov, const double *state_dev, PreintArgs *a) {
    char *Dv = nullptr;
    XR_HIP(hipHostGetDevicePointer((void **)&Dv, c->h_stage, 0));
    *a = PreintArgs{(const PreintJob *)(Dv + c->preint_o_jobs), (const double *)(Dv + c->preint_o_smp), (const double *)(Dv + c->preint_o_noise),
                    jac ? 1 : 0, cov ? 1 : 0, (double *)(Dv + c->preint_o_out), (int *)(Dv + c->preint_o_st),
                    (int *)(Dv + c->preint_o_st) + n_jobs, state_dev, n_jobs, {}};
    const PreintJob *hj = (const PreintJob *)(c->h_stage + c->preint_o_jobs);
    for (int k = 0; k < std::min(n_jobs, PI_HEAD); ++k) a->head[k] = hj[k];
    return XRHIP_OK;
}

static int preint_launch(xrhip_ba *c, int n_jobs, int jac, int cov, const double *state_dev) {
    if (c->group) {
        int rc = group_wait_launched(&c->rq_preint);   // (its argument block is about to be rewritten)
        if (rc) return rc;
    }
    int rc = preint_args(c, n_jobs, jac, cov, state_dev, &c->a_preint);
    if (rc) return rc;
    c->rq_preint.kind = GK_PREINT;
    c->rq_preint.owner = c;
    c->rq_preint.payload = &c->a_preint;
    if (c->group) {
        rc = group_submit(c->group, GQ_PREINT, &c->rq_preint);
        c->preint_rq = &c->rq_preint;
        c->preint_stream = group_stream(c->group, GQ_PREINT);
    } else {
        GroupRequest *one = &c->rq_preint;
        rc = launch_preint_batch(&one, 1, c->stream);
        c->preint_rq = nullptr;
        c->preint_stream = c->stream;
    }
    if (rc) return rc;
    c->preint_pending = n_jobs;
    c->preint_deferred = 0;
    return XRHIP_OK;
}

// The deferred batch of xrhip_ba_preintegrate_after_solve, as the argument set of a launch right behind the solve's kernel
// (state_dev: the solve's frame states on the device, final once that kernel has run).  *have = false: nothing was staged.
static int preint_fill_deferred(xrhip_ba *c, const xrhip_ba_problem *P, const double *state_dev, PreintArgs *out, bool *have) {
    *have = false;
    if (!c->preint_deferred) return XRHIP_OK;
    const int n_jobs = c->preint_deferred;
    const PreintJob *jobs = (const PreintJob *)(c->h_stage + c->preint_o_jobs);
    for (int k = 0; k < n_jobs; ++k)
        if (jobs[k].bias_frame < 0 || jobs[k].bias_frame >= P->n_frames) {
            c->preint_deferred = 0;
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_after_solve: bias frame is not a frame of the solve");
        }
    int rc = preint_args(c, n_jobs, c->preint_def_jac, c->preint_def_cov, state_dev, out);
    if (rc) return rc;
    *have = true;
    return XRHIP_OK;
}

// The deferred batch of xrhip_ba_preintegrate_after_solve, launched by xrhip_ba_solve on the paths that have returned the solve's
// states to the host already (P->frame_state): the biases go into the jobs by value.  (The single-launch solve launches it itself,
// behind its kernel: preint_fill_deferred.)
static int preint_launch_deferred(xrhip_ba *c, const xrhip_ba_problem *P, const double *state_dev) {
    if (!c->preint_deferred) return XRHIP_OK;
    const int n_jobs = c->preint_deferred;
    PreintJob *jobs = (PreintJob *)(c->h_stage + c->preint_o_jobs);
    for (int k = 0; k < n_jobs; ++k) {
        const int f = jobs[k].bias_frame;
        if (f < 0 || f >= P->n_frames) {
            c->preint_deferred = 0;
            return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_after_solve: bias frame is not a frame of the solve");
        }
        if (!state_dev) {
            for (int i = 0; i < 3; ++i) {
                jobs[k].bg[i] = P->frame_state[16 * f + 10 + i];
                jobs[k].ba[i] = P->frame_state[16 * f + 13 + i];
            }
            jobs[k].bias_frame = -1;
        }
    }
    return preint_launch(c, n_jobs, c->preint_def_jac, c->preint_def_cov, state_dev);
}

int xrhip_ba_preintegrate_begin(xrhip_ba *c, const double *samples, const int *sample_begin, const int *sample_count,
                                const double *t_end, const double *bg, const double *ba, int n_jobs,
                                const double *noise_cov36, int compute_jacobian, int compute_covariance) {
    if (!c || !samples || !sample_begin || !sample_count || !t_end || !bg || !ba || !noise_cov36 || n_jobs <= 0)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate: bad arguments");
    int rc = preint_stage(c, samples, sample_begin, sample_count, t_end, bg, ba, nullptr, n_jobs, noise_cov36);
    if (rc) return rc;
    return preint_launch(c, n_jobs, compute_jacobian, compute_covariance, nullptr);
}

int xrhip_ba_preintegrate_after_solve(xrhip_ba *c, const double *samples, const int *sample_begin, const int *sample_count,
                                      const double *t_end, const int *bias_frame, int n_jobs, const double *noise_cov36,
                                      int compute_jacobian, int compute_covariance) {
    if (!c || !samples || !sample_begin || !sample_count || !t_end || !bias_frame || !noise_cov36 || n_jobs <= 0)
        return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_after_solve: bad arguments");
    for (int k = 0; k < n_jobs; ++k)
        if (bias_frame[k] < 0) return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_after_solve: negative bias frame");
    int rc = preint_stage(c, samples, sample_begin, sample_count, t_end, nullptr, nullptr, bias_frame, n_jobs, noise_cov36);
    if (rc) return rc;
    c->preint_deferred = n_jobs;
    c->preint_def_jac = compute_jacobian ? 1 : 0;
    c->preint_def_cov = compute_covariance ? 1 : 0;
    return XRHIP_OK;
}

int xrhip_ba_preintegrate_cancel(xrhip_ba *c) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_cancel: null context");
    if (c->preint_pending) {   // its kernel writes the staging block: wait before anybody reuses it
        if (c->preint_rq) group_wait_launched(c->preint_rq);
        XR_HIP(hipStreamSynchronize(c->preint_stream ? c->preint_stream : c->stream));
        c->preint_pending = 0;
    }
    c->preint_deferred = 0;
    return XRHIP_OK;
}

int xrhip_ba_preintegrate_early(xrhip_ba *c, int job, double *out_delta11) {
    if (!c || !out_delta11 || job < 0) return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_early: bad arguments");
    if (!c->preint_pending || job >= c->preint_pending)
        return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate_early: no such job in flight");
    char *H = c->h_stage;
    const int n_jobs = c->preint_pending;
    volatile int *early = (volatile int *)(H + c->preint_o_st) + n_jobs;
    if (c->preint_rq) {
        const int rc = group_wait_launched(c->preint_rq);
        if (rc) return rc;
    }
    for (unsigned long spin = 1; early[job] == 0; ++spin)
        if ((spin & 0x3FFF) == 0) {
            const hipError_t q = hipStreamQuery(c->preint_stream ? c->preint_stream : c->stream);
            if (q == hipSuccess && early[job] == 0) return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate_early: kernel retired without publishing");
            if (q != hipSuccess && q != hipErrorNotReady) return xr_fail(XRHIP_EHIP, "xrhip_ba_preintegrate_early: stream error");
        }
    std::memcpy(out_delta11, H + c->preint_o_out + sizeof(double) * XRHIP_IMU_DIM * (size_t)job, sizeof(double) * 11);
    return XRHIP_OK;
}

int xrhip_ba_preintegrate_end(xrhip_ba *c, double *out) {
    if (!c || !out) return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate_end: bad arguments");
    if (c->preint_deferred) {
        c->preint_deferred = 0;
        return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate_end: the batch waits for a solve that never ran");
    }
    if (!c->preint_pending) return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate_end: nothing in flight");
    const int n_jobs = c->preint_pending;
    c->preint_pending = 0;
    char *H = c->h_stage;
    volatile int *st = (volatile int *)(H + c->preint_o_st);
    if (c->preint_rq) {   // grouped: the stream below is shared -- idle says nothing before the request has been launched
        const int rc = group_wait_launched(c->preint_rq);
        if (rc) return rc;
    }
    for (int k = 0; k < n_jobs; ++k)
        for (unsigned long spin = 1; st[k] == 0; ++spin)
            if ((spin & 0x3FFF) == 0) {
                const hipError_t q = hipStreamQuery(c->preint_stream ? c->preint_stream : c->stream);
                if (q == hipSuccess && st[k] == 0) return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate: kernel retired without publishing");
                if (q != hipSuccess && q != hipErrorNotReady) return xr_fail(XRHIP_EHIP, "xrhip_ba_preintegrate: stream error");
            }
    for (int k = 0; k < n_jobs; ++k)
        if (st[k] != 1) return xr_fail(XRHIP_ESTATE, "xrhip_ba_preintegrate: covariance is not positive definite");
    std::memcpy(out, H + c->preint_o_out, sizeof(double) * XRHIP_IMU_DIM * (size_t)n_jobs);
    return XRHIP_OK;
}

int xrhip_ba_preintegrate_batch(xrhip_ba *c, const double *samples, const int *sample_begin, const int *sample_count,
                                const double *t_end, const double *bg, const double *ba, int n_jobs,
                                const double *noise_cov36, int compute_jacobian, int compute_covariance, double *out) {
    if (!out) return xr_fail(XRHIP_EINVAL, "xrhip_ba_preintegrate: bad arguments");
    int rc = xrhip_ba_preintegrate_begin(c, samples, sample_begin, sample_count, t_end, bg, ba, n_jobs, noise_cov36,
                                         compute_jacobian, compute_covariance);
    if (rc) return rc;
    return xrhip_ba_preintegrate_end(c, out);
}

int xrhip_ba_preintegrate(xrhip_ba *c, const double *samples, int n, double t_end, const double *bg, const double *ba,
                          const double *noise_cov36, int compute_jacobian, int compute_covariance, double *out) {
    const int begin = 0;
    return xrhip_ba_preintegrate_batch(c, samples, &begin, &n, &t_end, bg, ba, 1, noise_cov36, compute_jacobian,
                                       compute_covariance, out);
}

}   // extern "C"
