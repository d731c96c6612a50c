// ba_chain.hip.h -- a whole solve in ONE launch, LDS-resident, for problems WITHOUT a free landmark and without a
// marginalisation prior: localize_newframe (one free frame against constant landmarks, sliding_window_tracker.cpp:119-143),
// refine_subwindow (a chain of free subframes tied by IMU factors, :370-465) and the initialiser's PnP (initializer.cpp:296-314).
//
// Why: these solves are tiny (15..90 unknowns, a few hundred observations, ~0.3 MFLOP per round) and there are one or
// two of them per frame.  The generic path (ba_kernels.hip.h) runs them through bodies written for window-sized problems --
// full 15F x 15F Hessian layout in global memory, one phase per global-memory round trip -- and measured 113-155 us
// per localize_newframe solve of ONE trust-region round (profiles/r02_kprof_v19.md): almost all of it latency between
// phases.  Here the unknowns are the `na` free frame dofs only; the Hessian (packed triangle), its scaled copy, all
// vectors, the frame states and the whitened IMU Jacobians live in LDS; the only global traffic is the observation
// list (read once per evaluation) and the per-observation Jacobian records between linearisation and block sums.
//
// The minimiser is the same code as everywhere else (TrialScalars, trial_begin, dogleg_point, trial_decide) and the
// arithmetic follows the generic bodies expression by expression (same factor order in every sum of the assembly,
// same thread -> element mapping in the block sums), so the iteration / accept / reject sequence and the states
// agree with the generic path and with the oracle exactly as before (tests/test_ba_gpu.py, golden snapshots).
// Trials whose dogleg point is the one just costed (Gauss-Newton step inside the radius, radius halved after a
// rejection: still inside) reuse that cost instead of evaluating the identical candidate again.
#pragma once
#include "ba_kernels.hip.h"

namespace xrhip {

constexpr int CHAIN_MAX_NA = 90;    // six free frames
constexpr int CHAIN_MAX_NI = 8;
constexpr int CHAIN_MAX_F = 64;
constexpr int CHAIN_MAX_OBS = 1024; // reprojection + rotation factors: four per thread
constexpr int CHAIN_MAX_FREE = 6;
// Four wavefronts, one per SIMD: the kernel holds ~480 registers per lane, so a fifth wavefront (two on one SIMD) would halve the
// budget and spill (measured in round 2: slower).
constexpr int CHAIN_THREADS = 256;

#ifdef XRHIP_KPROF
#define CPROF(slot)                                   \
    do {                                              \
        if (threadIdx.x == 0) {                       \
            const long long cp_n = wall_clock64();    \
            s_ctl.prof[slot] += cp_n - cp_t;          \
            cp_t = cp_n;                              \
        }                                             \
    } while (0)
#else
#define CPROF(slot) \
    do {            \
    } while (0)
#endif

struct ChainLayout {   // offsets in doubles into the dynamic LDS block
    int sx, cs, Hp, gp, A, scr, sp, D, gs, grad, gn, delta, gt, wJi, wJj, wr, bref, raw, Hv, oc, rec, xch, total;
};
__host__ __device__ __forceinline__ ChainLayout chain_layout(int F, int na, int NI, int nfree, int nobs) {
    ChainLayout L;
    int o = 0;
    auto take = [&](int n) {
        const int r = o;
        o += (n + 1) & ~1;
        return r;
    };
    L.sx = take(16 * F);
    L.cs = take(16 * F);
    L.Hp = take(na * (na + 1) / 2);
    L.gp = take(na);
    // the reduced system: at most 16 unknowns never leave registers; beyond that the tiled layout of dense_lds.hip.h + L^-1 rhs
    L.A = take(na <= CH_NB ? 2 : tl_doubles(na + 1) + 16 * tl_tile_rows(na + 1));
    L.scr = take(NI * IMU_SCR);   // raw IMU residuals / Jacobians: a region of its own, so that its zero pattern survives the rounds
    L.sp = take(na);
    L.D = take(na);
    L.gs = take(na);
    L.grad = take(na);
    L.gn = take(na);
    L.delta = take(na);
    L.gt = take(na);
    L.wJi = take(225 * NI);
    L.wJj = take(225 * NI);
    L.wr = take(16 * NI);
    L.bref = take(6 * NI);
    L.raw = take(15 * NI);
    L.Hv = take(28 * nfree);
    L.oc = take(nobs);   // per-factor costs of the reprojection / rotation factors (evaluated by whoever is free, summed in fixed order)
    L.rec = take(XRHIP_IMU_DIM * NI);   // the IMU records (pre-integrated measurement + sqrt_inv_cov): constant over the solve
    L.xch = take(IMU_XCH * NI);         // matrices handed from one wavefront to another inside a linearisation
    L.total = o;
    return L;
}

// (Tried: this workgroup pulling the staged problem from the pinned host arena itself instead of a kb_stage launch in
// front of it -- one compute unit reads the host link slower than kb_stage's 128 workgroups: localize_newframe
// 0.132 -> 0.139 ms per frame, profiles/r02_ab_variants.md.)
constexpr int CHAIN_VIS_TILE = 4 * 27 * 65;   // doubles of LDS behind the layout when `opts & CHAIN_OPT_TILE`: a [27][65] tile per wavefront
constexpr int CHAIN_OBS_CACHE = 16;        // doubles per reprojection factor behind that when `opts & CHAIN_OPT_CACHE`
enum { CHAIN_OPT_TILE = 1, CHAIN_OPT_CACHE = 2 };
__host__ __device__ __forceinline__ int chain_cache_stride(int M) { return (M + 1) & ~1; }

// A reprojection factor of these problems sees a constant landmark and -- mostly -- a constant reference frame: its tangent basis,
// the landmark in the reference rig and in the world do not change during the solve.  They are evaluated once (set-up) into an
// LDS table, SoA over the factors: rows 0-2 b1, 3-5 b2, 6-8 y_ref_center, 9-11 x, 12-14 z_tgt, 15 = tgt | ref << 8 | tgt free << 16
// | ref free << 17 (as a double).  An evaluation is then ~450 instead of ~1000 double-precision instructions per lane -- and the
// linearisation of the reprojection factors, not the IMU factor, was the longest stream of a round (profiles/r03_ab_variants.md).
__device__ __forceinline__ double obs_eval_cached(const BaDims &d, const double *oca, int Mp, int o, const double *state, const Ext &cam,
                                                  double sx, double sy, bool want_j, double *rec, const double *ftab, const int *slot_of,
                                                  const double *ctab) {
    const int inf = (int)oca[15 * Mp + o];
    const int ft = inf & 255, fr = (inf >> 8) & 255;
    const bool at = (inf >> 16) & 1, ar = (inf >> 17) & 1;
    if (!at && !ar) {
        if (want_j)
            for (int i = 0; i < OREC; ++i) rec[i] = 0.0;
        return 0.0;
    }
    ObsConst c;
    c.b1 = v3(oca[o], oca[Mp + o], oca[2 * Mp + o]);
    c.b2 = v3(oca[3 * Mp + o], oca[4 * Mp + o], oca[5 * Mp + o]);
    c.y_ref_center = v3(oca[6 * Mp + o], oca[7 * Mp + o], oca[8 * Mp + o]);
    c.x = v3(oca[9 * Mp + o], oca[10 * Mp + o], oca[11 * Mp + o]);
    const V3 zt = v3(oca[12 * Mp + o], oca[13 * Mp + o], oca[14 * Mp + o]);
    double r[2], Jt[12], Jr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Jr[i] = 0.0;
    if (!ar) {
        // free target, constant reference (every factor of localize_newframe, nearly every one of refine_subwindow): from the frame's
        // table (ftab: R^T and p of the free frames at the state being evaluated, frame_table) and the camera's (ctab)
        eval_reprojection_tgt(ftab + 12 * slot_of[ft], ctab, c, zt, sx, sy, r, want_j, Jt);
    } else {
        const FState st = load_state(state + 16 * ft), sr = load_state(state + 16 * fr);
        eval_reprojection_cached(st, sr, ar, c, zt, cam, sx, sy, r, want_j, Jt, Jr);
    }
    const double s = r[0] * r[0] + r[1] * r[1];
    if (want_j) {
        const double sc = d.robust ? sqrt(fmax(2.2250738585072014e-308, 1.0 / (1.0 + s))) : 1.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            rec[i] = at ? Jt[i] * sc : 0.0;
            rec[12 + i] = ar ? Jr[i] * sc : 0.0;
        }
        rec[24] = 0.0;
        rec[25] = 0.0;
        rec[26] = r[0] * sc;
        rec[27] = r[1] * sc;
    }
    return d.robust ? 0.5 * log(1.0 + s) : 0.5 * s;
}

// One workgroup per solve; a launch carries up to XB solves (blockIdx.z = entry: the solves of several sequences of an instance
// group, group.hip.h), each with its own staged problem, mailbox and options.  The dynamic LDS of the launch is the largest entry's.
// link_src / link_frame (xrhip_ba_solve_chained): frame `link_frame` of this problem starts from the 16 doubles at link_src -- where the
// solve launched in front of this one on the same stream left one of ITS frames (localize_newframe's result is refine_subwindow's
// starting point for the new frame) -- instead of from what was staged; link_frame < 0: none.
struct ChainArgs {
    const TinyArgs *args;
    int seq, max_rounds, opts;
    const double *link_src;
    int link_frame;
};
__global__ __launch_bounds__(CHAIN_THREADS) void kb_chain(Batch<ChainArgs> batch) {
    const TinyArgs *__restrict__ args = batch.e[blockIdx.z].args;
    const int seq = batch.e[blockIdx.z].seq, max_rounds = batch.e[blockIdx.z].max_rounds, opts = batch.e[blockIdx.z].opts;
    const double *link_src = batch.e[blockIdx.z].link_src;
    const int link_frame = batch.e[blockIdx.z].link_src ? batch.e[blockIdx.z].link_frame : -1;
    const BaDims &d = args->d;
    const BaPtrs &p = args->p;
    const Ext &cam = args->cam, &imu = args->imu;
    const double sx_ = args->sx, sy_ = args->sy;
    extern __shared__ double lds[];
    __shared__ double scratch[64];
    __shared__ double Dblk[CH_NB][CH_NB + 1];
    __shared__ double s_vis[4][CHAIN_MAX_FREE][27];
    __shared__ int s_fail;
    __shared__ int s_free[CHAIN_MAX_FREE], s_slot[CHAIN_MAX_F], s_nfree;
    __shared__ double s_ftab[2][CHAIN_MAX_FREE][12], s_ctab[12];   // frame_table of the free frames at x / at the candidate; ext_table
    __shared__ BaCtl s_ctl;
    const int tid = threadIdx.x, lane = tid & 63;
    // Who evaluates what.  A double-precision instruction costs its wavefront ~8 cycles of issue whatever the number of active
    // lanes (tools/latency.hip), so an IMU factor -- ~900 instructions for the residual, ~1300 more for the Jacobians, on ONE lane --
    // is the longest instruction stream of a round by far, and lanes of one wavefront run in lockstep: a wavefront that holds two
    // kinds of work runs them one after the other.  The streams of an IMU factor are therefore cut along their data dependencies
    // and dealt to the four wavefronts (lane k of every wavefront: its piece of factor k; ba_math.hip.h, imu_residual_rq ...):
    //   phase A   wavefront 3: the rotation residual (expmap, three quaternion products, logmap), then Jr^-1(rq), R(expmap(rq))^T and
    //             the rest of the residual
    //             wavefronts 0-2: the reprojection / rotation factors (stride 192), then the rq-free pieces of the IMU factor
    //   phase B   wavefronts 0, 1: the two triple products
    // The factors' costs go through LDS and are added up in the fixed order o = t, t + 256, ... of the generic bodies, so the sums
    // do not depend on who evaluated what.
    const bool split = d.NI > 0;
    const bool imu_lane = lane < d.NI;          // in every wavefront
    const int ostride = split ? 192 : 256;
    const int otid = (!split || tid < 192) ? tid : (1 << 30);
    const int wtid = tid, wave = tid >> 6;
    constexpr bool worker = true;
    constexpr int nt = 256;
    const int F = d.F, n = d.n, na = d.na, NI = d.NI, M = d.M, MR = d.MR;
#ifdef XRHIP_KPROF
    long long cp_t = wall_clock64();
    const long long cp_t0 = cp_t;
#endif

    // free frames in index order (F <= 64: one lane per frame, the slot is the number of free frames before it) and the
    // control block, one word per thread -- a single thread walking both lists was 4 us of every solve
    if (tid < 64) {
        const bool fr = tid < F && p.fix[tid < F ? tid : 0] != 3;
        const unsigned long long m = __ballot(fr);
        const int slot = __popcll(m & ((1ull << tid) - 1ull));
        if (tid < F) s_slot[tid] = (fr && slot < CHAIN_MAX_FREE) ? slot : -1;
        if (fr && slot < CHAIN_MAX_FREE) s_free[slot] = tid;
        if (tid == 0) s_nfree = min(__popcll(m), CHAIN_MAX_FREE);
    } else {
        constexpr unsigned ctl_words = sizeof(BaCtl) / sizeof(long long);
        const long long *src = reinterpret_cast<const long long *>(static_cast<BaCtl *>(p.ctl));
        long long *dst = reinterpret_cast<long long *>(&s_ctl);
        for (unsigned i = tid - 64; i < ctl_words; i += nt - 64) dst[i] = src[i];
    }
    __syncthreads();
    const int nfree = s_nfree;
    const ChainLayout Lo = chain_layout(F, na, NI, nfree, M + MR);
    double *const X = lds + Lo.sx, *const CS = lds + Lo.cs, *const Hp = lds + Lo.Hp, *const gp = lds + Lo.gp;
    double *const A = lds + Lo.A, *const sp = lds + Lo.sp, *const Dg = lds + Lo.D, *const gs = lds + Lo.gs;
    double *const grad = lds + Lo.grad, *const gn = lds + Lo.gn, *const delta = lds + Lo.delta, *const gt = lds + Lo.gt;
    double *const wJi = lds + Lo.wJi, *const wJj = lds + Lo.wJj, *const wr = lds + Lo.wr, *const bref = lds + Lo.bref;
    double *const raw = lds + Lo.raw, *const Hv = lds + Lo.Hv, *const oc = lds + Lo.oc, *const recs = lds + Lo.rec, *const xch = lds + Lo.xch;
    double *const scr = lds + Lo.scr;   // [NI][IMU_SCR]
    const bool vis_tile = (opts & CHAIN_OPT_TILE) != 0, use_cache = (opts & CHAIN_OPT_CACHE) != 0;
    const int Mp = chain_cache_stride(M);
    double *const oca = lds + Lo.total + (vis_tile ? CHAIN_VIS_TILE : 0);   // [16][Mp]
    BaCtl *const c = &s_ctl;

    // this thread's slots of the full 15F layout (element a = tid + 256 m, like the strided loops of the generic bodies):
    // act[m] = index of that dof among the free ones, or -1
    int act[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int a = tid + nt * m;
        act[m] = (worker && a < n) ? p.act_inv[a] : -1;
    }
    for (int e = wtid; e < 16 * F; e += nt) X[e] = (e >> 4) == link_frame ? link_src[e & 15] : p.state[e];
    // (the bias reference of an IMU factor is the bias of its first frame as the solve starts: the linked frame's comes with its state)
    for (int e = wtid; e < 6 * NI; e += nt) bref[e] = p.imu_i[e / 6] == link_frame ? link_src[10 + e % 6] : p.bias_ref[e];
    for (int e = wtid; e < XRHIP_IMU_DIM * NI; e += nt) recs[e] = p.imu_data[e];
    for (int e = wtid; e < NI * IMU_SCR; e += nt) scr[e] = 0.0;   // the blocks a linearisation writes are the same every round
    __syncthreads();
    if (tid < nfree) frame_table(load_state(X + 16 * s_free[tid]), s_ftab[0][tid]);
    else if (tid == 64) ext_table(cam, s_ctab);
    if (use_cache) {
        for (int o = wtid; o < M; o += nt) {
            const int ft = p.obs_tgt[o], fr = p.obs_ref[o];
            const bool at = pose_free(p.fix[ft]), ar = pose_free(p.fix[fr]);
            const V3 zt = v3(p.obs_zt[3 * o], p.obs_zt[3 * o + 1], p.obs_zt[3 * o + 2]);
            if (at || ar) {
                const V3 zr = v3(p.obs_zr[3 * o], p.obs_zr[3 * o + 1], p.obs_zr[3 * o + 2]);
                const ObsConst c = reprojection_constants(load_state(X + 16 * fr), p.depth[p.obs_lm[o]], zt, zr, cam);
                oca[o] = c.b1.x; oca[Mp + o] = c.b1.y; oca[2 * Mp + o] = c.b1.z;
                oca[3 * Mp + o] = c.b2.x; oca[4 * Mp + o] = c.b2.y; oca[5 * Mp + o] = c.b2.z;
                oca[6 * Mp + o] = c.y_ref_center.x; oca[7 * Mp + o] = c.y_ref_center.y; oca[8 * Mp + o] = c.y_ref_center.z;
                oca[9 * Mp + o] = c.x.x; oca[10 * Mp + o] = c.x.y; oca[11 * Mp + o] = c.x.z;
                oca[12 * Mp + o] = zt.x; oca[13 * Mp + o] = zt.y; oca[14 * Mp + o] = zt.z;
            }
            oca[15 * Mp + o] = (double)(ft | (fr << 8) | ((int)at << 16) | ((int)ar << 17));
        }
    }
    __syncthreads();   // the table of constants, the frames' tables
    CPROF(0);   // set-up: control block, states, index slots

    bool relin = true;
    int mode = 1, st = ST_RUNNING;
    for (int round = 0; round < max_rounds; ++round) {
        if (relin) {
            // ---------------- linearisation (the schedule at the top of the kernel)
            // phase A
#ifdef XRHIP_KPROF
            const long long pa_t0 = wall_clock64();
#endif
            for (int o = otid; o < M; o += ostride) {
                double rec[OREC];
                const double co = use_cache ? obs_eval_cached(d, oca, Mp, o, X, cam, sx_, sy_, true, rec, &s_ftab[0][0][0], s_slot, s_ctab)
                                            : obs_eval(d, p, o, X, p.depth, cam, sx_, sy_, true, rec);
                oc[o] = co;
#pragma unroll
                for (int i = 0; i < OREC; ++i) p.orec[(size_t)o * OREC + i] = rec[i];
            }
            for (int o = otid; o < MR; o += ostride) {
                double rec[RREC];
                const double co = rot_eval(d, p, o, X, cam, sx_, sy_, true, rec);
                oc[M + o] = co;
#pragma unroll
                for (int i = 0; i < RREC; ++i) p.rrec[(size_t)o * RREC + i] = rec[i];
            }
#ifdef XRHIP_KPROF
            const long long pa_t1 = wall_clock64();
            if (tid == 0) s_ctl.prof[21] += pa_t1 - pa_t0;   // wavefront 0: its reprojection / rotation factors
#endif
            bool imu_on = false;
            int nfi = 0, nfj = 0;
            if (imu_lane) {
                const int k = lane, fi = p.imu_i[k], fj = p.imu_j[k];
                nfi = p.fix[fi] != 3;
                nfj = p.fix[fj] != 3;
                imu_on = nfi || nfj;
                if (imu_on) {
                    double *rw = scr + k * IMU_SCR;
                    const FState si = load_state(X + 16 * fi), sj = load_state(X + 16 * fj);
                    const ImuRec pre = load_imu(recs + k * XRHIP_IMU_DIM);
                    const V3 bg0 = v3(bref[6 * k], bref[6 * k + 1], bref[6 * k + 2]);
                    const V3 ba0 = v3(bref[6 * k + 3], bref[6 * k + 4], bref[6 * k + 5]);
                    if (wave == 3) {
                        // the long pole: the rotation residual, then -- without waiting for anybody -- what only needs it
                        const V3 rq = imu_residual_rq(si, sj, pre, bg0, imu);
                        rw[0] = rq.x;
                        rw[1] = rq.y;
                        rw[2] = rq.z;
                        double *xk = xch + k * IMU_XCH;
                        store33(xk + 45, imu_jac_jrinv(rq));
                        if (nfi) store33(xk + 36, imu_jac_B(rq));
                        imu_residual_rest(si, sj, pre, bg0, ba0, imu, rw);   // this wavefront has no reprojection factors
                    } else if (wave == 0) {
                        imu_raw_jacobians_part(2, si, sj, pre, bg0, ba0, imu, v3(0, 0, 0), rw + 15, rw + 240, nfi, nfj);
                    } else if (wave == 1) {
                        imu_jac_pre(si, sj, pre, bg0, imu, xch + k * IMU_XCH, nfi, nfj);
                    } else {
                        imu_raw_jacobians_part(3, si, sj, pre, bg0, ba0, imu, v3(0, 0, 0), rw + 15, rw + 240, nfi, nfj);
                    }
                }
            }
#ifdef XRHIP_KPROF
            if (tid == 192) atomicAdd((unsigned long long *)&s_ctl.prof[22], (unsigned long long)(wall_clock64() - pa_t1));   // wavefront 3: rq chain + Jr^-1, B
            if (tid == 0) atomicAdd((unsigned long long *)&s_ctl.prof[23], (unsigned long long)(wall_clock64() - pa_t1));     // wavefront 0: its rq-free piece
#endif
            __syncthreads();
            CPROF(18);  // lin: phase A (reprojection factors; rotation residual, Jr^-1, B; rq-free pieces)
            // ---------------- reprojection blocks of the free frames: the (f, f) pair list, upper triangle + gradient = 27 sums per frame.
            // Every participating wavefront reduces ITS lanes' partial sums through a tile of its own (a store of 27 columns, 54 lanes add
            // half a row each) and goes on to the next frame at once (rounds 2-5: a 256-wide tile and three barriers per free frame).
            // With IMU factors the blocks are wavefronts 2 and 3's, BESIDE phase B (the IMU Jacobians' products, wavefronts 0 and 1): they
            // read only what phase A left in the factor records.
            auto vis_blocks = [&](int idx, int stride) __attribute__((always_inline)) {
                for (int s = 0; s < nfree; ++s) {
                    const int f = s_free[s], pair = f * F + f;
                    const int s0 = p.pair_start[pair], s1 = p.pair_start[pair + 1];
                    double acc[27];
#pragma unroll
                    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
                    for (int it = s0 + idx; it < s1; it += stride) {
                        const int code = p.pair_items[it];
                        const double *rec = p.orec + (size_t)(code >> 1) * OREC + ((code & 1) ? 12 : 0);
                        double j[12];
#pragma unroll
                        for (int i = 0; i < 12; ++i) j[i] = rec[i];
                        const double *rr = p.orec + (size_t)(code >> 1) * OREC + 26;
                        const double r0 = rr[0], r1 = rr[1];
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) {
#pragma unroll
                            for (int b = a; b < 6; ++b) acc[e++] += j[a] * j[b] + j[6 + a] * j[6 + b];
                            acc[21 + a] += j[a] * r0 + j[6 + a] * r1;
                        }
                    }
                    if (vis_tile) {
                        double *tw = lds + Lo.total + wave * (27 * 65);
#pragma unroll
                        for (int i = 0; i < 27; ++i) tw[i * 65 + lane] = acc[i];
                        wave_sync();
                        const int row = lane < 27 ? lane : lane - 27;   // lanes 0-26: entries 0-31 of row `lane`; lanes 27-53: entries 32-63
                        double s2 = 0.0;
                        if (lane < 54) {
                            const double *rp = tw + row * 65 + (lane < 27 ? 0 : 32);
#pragma unroll
                            for (int i = 0; i < 32; ++i) s2 += rp[i];
                        }
                        const double hi = __shfl(s2, (lane + 27) & 63);   // every lane active (a ds_bpermute reads inactive lanes as zero)
                        if (lane < 27) s_vis[wave][s][lane] = s2 + hi;
                        wave_sync();   // the tile is free for the next frame
                    } else {
#pragma unroll
                        for (int i = 0; i < 27; ++i) acc[i] = wave_sum(acc[i]);
                        if (lane == 0)
#pragma unroll
                            for (int i = 0; i < 27; ++i) s_vis[wave][s][i] = acc[i];
                    }
                }
            };
            if (split) {
                // phase B: the products (wavefronts 0, 1) beside the reprojection blocks (wavefronts 2, 3)
                if (wave >= 2) {
                    vis_blocks(tid - 128, 128);
                } else if (imu_on) {
                    double *rw = scr + lane * IMU_SCR;
                    const double *xk = xch + lane * IMU_XCH;
                    if (wave == 0) imu_jac_finish0(load33(xk + 45), xk, rw + 15, rw + 240, nfi, nfj);
                    else if (wave == 1)
                        imu_jac_finish1(load33(xk + 45), load33(xk + 36), xk, load33(recs + lane * XRHIP_IMU_DIM + 11), rw + 15, nfi);
                }
            } else {
                vis_blocks(tid, nt);
            }
            __syncthreads();
            CPROF(1);   // linearisation of the factors; reprojection blocks
            // the wavefronts' block sums, in wavefront order (beside the whitening below; the assembly reads them behind its barrier)
            for (int e = tid; e < 27 * nfree; e += nt) {
                const int s = e / 27, q = e - 27 * s;
                Hv[28 * s + q] = split ? s_vis[2][s][q] + s_vis[3][s][q] : (s_vis[0][s][q] + s_vis[1][s][q]) + (s_vis[2][s][q] + s_vis[3][s][q]);
            }
            // ---------------- whitening of the IMU factors (lin_imu_block's second half): the 225 entries of a factor's two
            // Jacobians over the whole workgroup, the residual on the first lanes of wavefront 0; sqrt_inv_cov from LDS
            for (int k = 0; k < NI; ++k) {
                const double *rw = scr + k * IMU_SCR, *Ji = rw + 15, *Jj = rw + 240;
                const int fi = p.imu_i[k], fj = p.imu_j[k];
                const bool active = !(p.fix[fi] == 3 && p.fix[fj] == 3);
                const double *S = recs + k * XRHIP_IMU_DIM + 56;
                if (wave == (k & 3)) {
                    double cost = 0.0;
                    if (lane < 15) {
                        double s = 0;
                        if (active)
                            for (int j = 0; j < 15; ++j) s += S[15 * lane + j] * rw[j];
                        wr[16 * k + lane] = s;
                        cost = 0.5 * s * s;
                    }
                    cost = wave_sum(cost);
                    if (lane == 0) wr[16 * k + 15] = cost;
                }
                if (wtid < 225) {
                    const int e = wtid, i = e / 15, cc = e - 15 * i;
                    double a = 0, b = 0;
                    if (active) {
                        const bool col_i = cc < 6 ? pose_free(p.fix[fi]) : motion_free(p.fix[fi]);
                        const bool col_j = cc < 6 ? pose_free(p.fix[fj]) : motion_free(p.fix[fj]);
                        if (col_i)
                            for (int j = 0; j < 15; ++j) a += S[15 * i + j] * Ji[15 * j + cc];
                        if (col_j)
                            for (int j = 0; j < 15; ++j) b += S[15 * i + j] * Jj[15 * j + cc];
                    }
                    wJi[225 * k + e] = a;
                    wJj[225 * k + e] = b;
                }
            }
            __syncthreads();
            CPROF(2);   // IMU whitening, block sums
            // ---------------- assembly of the free x free entries (packed lower triangle) and of the gradient:
            // reprojection block, rotation priors, the IMU factor ending at the frame, the one starting at it --
            // assemble_item's order
            for (int e = wtid; e < na * (na + 1) / 2 + na; e += nt) {
                int i, j;
                const bool want_g = e >= na * (na + 1) / 2;
                if (want_g) {
                    i = e - na * (na + 1) / 2;
                    j = 0;
                } else {   // e = i (i + 1) / 2 + j
                    i = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);   // e < 2^12: exact enough, corrected below
                    while (i * (i + 1) / 2 > e) --i;
                    while ((i + 1) * (i + 2) / 2 <= e) ++i;
                    j = e - i * (i + 1) / 2;
                }
                const int a = p.act_idx[i], b = p.act_idx[j];
                const int fa = a / 15, ka = a - 15 * fa, fb = b / 15, kb = b - 15 * fb;
                double h = 0.0, g = 0.0;
                if (ka < 6) {
                    const double *hv = Hv + 28 * s_slot[fa];
                    if (!want_g && kb < 6 && fb == fa) {
                        const int lo = ka < kb ? ka : kb, hi = ka < kb ? kb : ka;
                        h += hv[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
                    }
                    if (want_g) g += hv[21 + ka];
                    if (ka < 3) {
                        const int s = p.rotf_start[fa], t = p.rotf_start[fa + 1];
                        for (int it = s; it < t; ++it) {
                            const double *rec = p.rrec + (size_t)p.rotf_items[it] * RREC;
                            if (!want_g && fb == fa && kb < 3) h += rec[ka] * rec[kb] + rec[3 + ka] * rec[3 + kb];
                            if (want_g) g += rec[ka] * rec[6] + rec[3 + ka] * rec[7];
                        }
                    }
                }
                for (int side = 0; side < 2; ++side) {
                    const int k = p.imuf[2 * fa + side];
                    if (k < 0) continue;
                    const double *Ja = (side == 0 ? wJj : wJi) + 225 * k;
                    const int fi = p.imu_i[k], fj = p.imu_j[k];
                    if (want_g) {
                        const double *r = wr + 16 * k;
                        for (int q = 0; q < 15; ++q) g += Ja[15 * q + ka] * r[q];
                    } else if (fb == fi || fb == fj) {
                        const double *Jb = (fb == fj ? wJj : wJi) + 225 * k;
                        for (int q = 0; q < 15; ++q) h += Ja[15 * q + ka] * Jb[15 * q + kb];
                    }
                }
                if (want_g) gp[i] = g;
                else Hp[e] = h;
            }
            CPROF(4);   // assembly
            // ---------------- total cost (sum_cost_block: observations, rotation factors, IMU factors k = tid, ...)
            double cost_part = 0.0;   // this thread's share of the total cost, in sum_cost_block's order
            for (int o = wtid; o < M; o += nt) cost_part += oc[o];
            for (int o = wtid; o < MR; o += nt) cost_part += oc[M + o];
            for (int k = wtid; k < NI; k += nt) cost_part += wr[16 * k + 15];
            const double ctot = block_sum(cost_part, scratch);
            if (tid == 0) {
                c->x_cost = ctot + 0.0;   // + the prior's cost: there is none
                if (c->first) {
                    c->initial_cost = c->x_cost;
                    c->minimum_cost = c->x_cost;
                }
            }
            __syncthreads();
            CPROF(5);   // cost
        }
        // ---------------- gradient max-norm |x - Plus(x, -g)|_inf over the free frames (gradmax_block): an exponential map on
        // a handful of lanes, ~3 us of which nothing below needs before the trial logic -- wavefront 1 computes it while
        // wavefront 0 factors the first diagonal block of the reduced system (chol_blocked's `side`)
        const bool want_gmax = relin;
        auto gradmax_side = [&]() __attribute__((always_inline)) {
            if (!want_gmax) return;
            double mx = 0;
            if (lane < nfree) {
                const int f = s_free[lane];
                double neg[15], out[16];
                for (int k = 0; k < 15; ++k) {
                    const int i = p.act_inv[15 * f + k];
                    neg[k] = i >= 0 ? -gp[i] : -0.0;
                }
                const double *s = X + 16 * f;
                state_plus(s, neg, pose_free(p.fix[f]), motion_free(p.fix[f]), out);
                for (int k = 0; k < 16; ++k) mx = fmax(mx, fabs(s[k] - out[k]));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
            if (lane == 0) c->gmax = mx;   // nfree <= 6 lanes of this wavefront
        };
        const double mu = c->mu;
        const int first_lin = c->first;   // read before anybody clears it below
        bool lin_ok = false;
        if (na <= CH_NB) {
            // ---------------- at most 16 unknowns (localize_newframe, PnP): the whole dense algebra of the round on ONE wavefront,
            // lane = unknown, no workgroup barrier inside -- preparation, reduced system, Q(g~, g~), Cholesky with the right-hand side
            // riding along, back-substitution, Gauss-Newton step and the six dogleg sums.  As four wavefronts with a block-wide
            // reduction and a barrier per step these five phases were 12.6 us of a 36 us round (in-kernel timers), almost all of it
            // waiting.  Beside it: wavefront 1 the gradient max-norm, wavefront 2 the bias-reference refresh and |x|.
            __syncthreads();   // Hp, gp, the cost of the linearisation (or, on a re-solve, the previous trial phase) are complete
            if (wave == 0) {
                const int i = lane;
                const bool on = i < na;
                const double h = on ? Hp[tri_idx(i, i)] : 1.0;
                double spi = on ? sp[i] : 0.0;
                if (first_lin) spi = on ? 1.0 / (1.0 + sqrt(h)) : 0.0;
                const double Dv = sqrt(fmin(fmax(spi * spi * h, 1e-6), 1e32));
                const double gsv = on ? spi * gp[i] : 0.0;
                const double gtv = on ? spi * (gsv / (Dv * Dv)) : 0.0;
                // The reduced system never leaves registers (round 6): lane i < 16 forms row i of S = sp H sp + mu D^2, lanes 16-31 carry
                // the unit rows, lane 32 the right-hand side; diag16_pivot_pairs leaves column r of L^-1 in lane 16 + r and L^-1 rhs in
                // lane 32, and x = L^-T (L^-1 rhs) is sixteen broadcast-multiply-adds.  (Rounds 2-5: the rows went to LDS, the block
                // routine read them back and wrote L, a 16-step back-substitution read L again.)
                double Xr[CH_NB];
                double t_i = 0.0;   // (H g~)_i
                const double yi = on ? gp[i] * spi : 0.0;
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) {
                    const double spk = lane_bcast(spi, k), gtk = lane_bcast(gtv, k), yk = lane_bcast(yi, k);
                    double v = (k == lane) ? 1.0 : 0.0;   // rows >= na of the block, and (lanes 16-31) the unit rows
                    if (on && k < na) {
                        const double hik = Hp[k <= i ? tri_idx(i, k) : tri_idx(k, i)];
                        t_i += hik * gtk;
                        v = hik * (spi * spk);
                        if (k == i) v += mu * Dv * Dv;
                    }
                    if (lane >= CH_NB) v = (lane == 2 * CH_NB) ? yk : ((k == lane - CH_NB) ? 1.0 : 0.0);
                    Xr[k] = v;
                }
                if (on) {
                    if (first_lin) sp[i] = spi;
                    Dg[i] = Dv;
                    gs[i] = gsv;
                }
                double qgg = gtv * t_i;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) qgg += __shfl_xor(qgg, off);
                qgg = lane_bcast(qgg, 0);
                bool ok = diag16_pivot_pairs(Xr, na, lane);
                double xs = 0.0;   // lane 16 + r: x_r = sum_c Linv[c][r] (L^-1 rhs)_c
#pragma unroll
                for (int cc = 0; cc < CH_NB; ++cc) xs = fma(Xr[cc], lane_bcast(Xr[cc], 2 * CH_NB), xs);
                // (the shuffle runs with every lane active: a ds_bpermute under `on` would read the inactive lanes 16-31 as zero)
                const double xs_up = __shfl(xs, (lane + CH_NB) & 63);
                const double ya = on ? xs_up : 0.0;
                if (__ballot(on && !isfinite(ya)) != 0ull) ok = false;
                const double gnv = on ? -Dv * ya : 0.0, grv = on ? gsv / Dv : 0.0;
                if (on) {
                    gn[i] = gnv;
                    grad[i] = grv;
                }
                // the six sums of the dogleg scalars over the unknowns (16 lanes: four shuffle stages each, interleaved)
                double r6[6] = {grv * grv, on ? (gnv / Dv) * gsv : 0.0, gnv * gnv, grv * grv, gnv * gnv, grv * gnv};
#pragma unroll
                for (int off = 8; off > 0; off >>= 1)
#pragma unroll
                    for (int q = 0; q < 6; ++q) r6[q] += __shfl_xor(r6[q], off);
                if (lane == 0) {
                    if (ok) {
                        c->alpha = r6[0] / qgg;
                        c->q_gg = qgg;
                        c->q_gn = -r6[0] - mu * r6[1];
                        c->q_nn = -r6[1] - mu * r6[2];
                    }
                    c->gnorm = ok ? sqrt(r6[3]) : 0.0;
                    c->gn_norm = ok ? sqrt(r6[4]) : 0.0;
                    c->gd = ok ? r6[5] : 0.0;
                    c->linear_ok = ok ? 1 : 0;
                }
            } else if (wave == 1) {
                gradmax_side();
            } else if (wave == 2 && mode == 1) {
                // FinalizeIterationAndCheckIfMinimizerCanContinue of the iteration that produced this x: the IMU factors read their
                // bias reference from the user state, refreshed by the StateUpdatingCallback; |x| over the free blocks
                if (lane < NI) {
                    const double *sti = X + 16 * p.imu_i[lane];
                    for (int q = 0; q < 6; ++q) bref[6 * lane + q] = sti[10 + q];
                }
                double s2 = 0;
                for (int f = lane; f < F; f += 64) {
                    const double *x = X + 16 * f;
                    if (pose_free(p.fix[f]))
                        for (int k = 0; k < 7; ++k) s2 += x[k] * x[k];
                    if (motion_free(p.fix[f]))
                        for (int k = 7; k < 16; ++k) s2 += x[k] * x[k];
                }
                s2 = wave_sum(s2);
                if (lane == 0) {
                    c->x_norm = sqrt(s2);
                    c->first = 0;
                }
            }
            __syncthreads();
            lin_ok = c->linear_ok != 0;
        } else {
            // -------------------- preparation (prepare_block): Jacobi scales at the first linearisation, dogleg diagonal
            for (int i = wtid; i < na; i += nt) {
                const double h = Hp[i * (i + 1) / 2 + i];
                if (c->first) sp[i] = 1.0 / (1.0 + sqrt(h));
                const double s = sp[i];
                const double Dv = sqrt(fmin(fmax(s * s * h, 1e-6), 1e32));
                Dg[i] = Dv;
                const double gsv = s * gp[i];
                gs[i] = gsv;
                gt[i] = s * (gsv / (Dv * Dv));
            }
            __syncthreads();
            // -------------------- reduced system S = sp H sp + mu D^2 in the tiled layout (the rhs as row na) and Q(g~, g~)
            const int nrows = na + 1, Tt = tl_tile_rows(nrows);
            double *yv = A + tl_doubles(nrows);   // [16 Tt]
            tl_clear(A, na, nrows);
            __syncthreads();
            for (int e = wtid; e < na * (na + 1) / 2; e += nt) {
                int i = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
                while (i * (i + 1) / 2 > e) --i;
                while ((i + 1) * (i + 2) / 2 <= e) ++i;
                const int j = e - i * (i + 1) / 2;
                double v = Hp[e] * (sp[i] * sp[j]);
                if (i == j) v += mu * Dg[i] * Dg[i];
                A[tl_idx(i, j)] = v;
            }
            for (int i = wtid; i < na; i += nt) A[tl_idx(na, i)] = gp[i] * sp[i];
            double qacc = 0;
            for (int i = wave; i < na; i += 4) {
                double t = 0;
                for (int j = lane; j < na; j += 64) t += Hp[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * gt[j];
                qacc += gt[i] * t;
            }
            const double qgg = block_sum(qacc, scratch);   // (block_sum ends with a barrier: the tiles are complete)
            CPROF(6);   // preparation, reduced system, Q(g~, g~)
            // -------------------- Cholesky + substitution (solve_block)
            lin_ok = tl_chol(A, na, nrows, &Dblk[0][0], &s_fail, nullptr, gradmax_side);
            CPROF(7);   // Cholesky
            if (lin_ok) {
                for (int i = wtid; i < 16 * Tt; i += nt) yv[i] = i < na ? A[tl_idx(na, i)] : 0.0;   // L^-1 rhs; zero beyond n (tl_trsv_t)
                __syncthreads();
                tl_trsv_t(A, na, yv);
                int bad = 0;
                for (int i = wtid; i < na; i += nt) {
                    const double ya = yv[i];
                    gn[i] = -Dg[i] * ya;
                    grad[i] = gs[i] / Dg[i];
                    if (!isfinite(ya)) bad = 1;
                }
                if (bad) atomicExch(&s_fail, 1);
                __syncthreads();
                if (s_fail) lin_ok = false;
            }
            __syncthreads();
            if (lin_ok) {
                double r3[3] = {0, 0, 0};   // |grad|^2, n~ . gs, |gn|^2   (a = tid, tid + 256, ... of the full layout)
    #pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int i = act[m];
                    if (i >= 0) {
                        const double g = grad[i], nn = gn[i];
                        r3[0] += g * g;
                        r3[1] += (nn / Dg[i]) * gs[i];
                        r3[2] += nn * nn;
                    }
                }
                block_sum_n<3>(r3, scratch);
                if (tid == 0) {
                    c->alpha = r3[0] / qgg;
                    c->q_gg = qgg;
                    c->q_gn = -r3[0] - mu * r3[1];
                    c->q_nn = -r3[1] - mu * r3[2];
                }
            }
            if (tid == 0) c->linear_ok = lin_ok ? 1 : 0;
            __syncthreads();
            CPROF(8);   // substitution, Gauss-Newton step, dogleg scalars
            // -------------------- trials (try_block)
            if (mode == 1) {
                // FinalizeIterationAndCheckIfMinimizerCanContinue of the iteration that produced this x: the IMU factors
                // read their bias reference from the user state, refreshed by the StateUpdatingCallback
                for (int k = wtid; k < NI; k += nt) {
                    const double *sti = X + 16 * p.imu_i[k];
                    for (int i = 0; i < 6; ++i) bref[6 * k + i] = sti[10 + i];
                }
                double s2 = 0;
                for (int f = wtid; f < F; f += nt) {
                    const double *x = X + 16 * f;
                    if (pose_free(p.fix[f]))
                        for (int k = 0; k < 7; ++k) s2 += x[k] * x[k];
                    if (motion_free(p.fix[f]))
                        for (int k = 7; k < 16; ++k) s2 += x[k] * x[k];
                }
                s2 = block_sum(s2, scratch);
                if (tid == 0) {
                    c->x_norm = sqrt(s2);
                    c->first = 0;
                }
                __syncthreads();
            }
            {
                double r3[3] = {0, 0, 0};
    #pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int i = act[m];
                    if (i >= 0 && lin_ok) {
                        r3[0] += grad[i] * grad[i];
                        r3[1] += gn[i] * gn[i];
                        r3[2] += grad[i] * gn[i];
                    }
                }
                block_sum_n<3>(r3, scratch);
                if (tid == 0) {
                    c->gnorm = sqrt(r3[0]);
                    c->gn_norm = sqrt(r3[1]);
                    c->gd = r3[2];
                }
                __syncthreads();
            }
        }
        CPROF(9);   // start of the trial phase (user-state refresh, norms)
        TrialScalars t;
        trial_load(c, t);
        bool check_gradient = (mode == 1), skip_finalize = (mode == 3);
        bool have_prev = false;
        double prev_ca = 0, prev_cb = 0, prev_mcc = 0, prev_cost = 0, prev_dn2 = 0, prev_sn = 0;
        int accepted = 0;
        while (t.status == ST_RUNNING) {
            trial_begin(t, skip_finalize, check_gradient);
            skip_finalize = false;
            check_gradient = false;
            if (t.status != ST_RUNNING) break;
            double ca, cb, step_norm;
            dogleg_point(t, t.radius, ca, cb, step_norm);
            double mcc, cost, dn2;
            if (have_prev && ca == prev_ca && cb == prev_cb && step_norm >= 0.0) {
                // the same dogleg point as the trial just rejected (not the interpolated case, whose coefficients depend
                // on the radius): same candidate, same sums
                mcc = prev_mcc;
                cost = prev_cost;
                dn2 = prev_dn2;
                step_norm = prev_sn;
            } else {
                double red2[2] = {0, 0};   // |step|^2 (D-scaled), step . gs
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int i = act[m];
                    if (i >= 0) {
                        const double v = ca * grad[i] + cb * gn[i];
                        red2[0] += v * v;
                        const double stv = v / Dg[i];
                        red2[1] += stv * gs[i];
                        delta[i] = stv * sp[i];
                    }
                }
                block_sum_n<2>(red2, scratch);
                if (step_norm < 0) step_norm = sqrt(red2[0]);
                mcc = dogleg_model_change(t, ca, cb, red2[1]);
                __syncthreads();
                CPROF(13);  // trial: dogleg point, step sums
                // candidate states (every frame, like the generic path: constant frames are copies)
                for (int f = wtid; f < F; f += nt) {
                    double d15[15];
                    for (int k = 0; k < 15; ++k) {
                        const int i = p.act_inv[15 * f + k];
                        d15[k] = i >= 0 ? delta[i] : 0.0;
                    }
                    state_plus(X + 16 * f, d15, pose_free(p.fix[f]), motion_free(p.fix[f]), CS + 16 * f);
                    if (s_slot[f] >= 0) frame_table(load_state(CS + 16 * f), s_ftab[1][s_slot[f]]);   // (this thread's own stores)
                }
                __syncthreads();
                CPROF(14);  // trial: candidate states
                double red[2] = {0, 0};   // cost, |x - candidate|^2
                for (int o = otid; o < M; o += ostride)
                    oc[o] = use_cache ? obs_eval_cached(d, oca, Mp, o, CS, cam, sx_, sy_, false, nullptr, &s_ftab[1][0][0], s_slot, s_ctab)
                                      : obs_eval(d, p, o, CS, p.depth, cam, sx_, sy_, false, nullptr);
                for (int o = otid; o < MR; o += ostride) oc[M + o] = rot_eval(d, p, o, CS, cam, sx_, sy_, false, nullptr);
                if (imu_lane && wave >= 2) {   // lane k of wavefront 3: the rotation residual of factor k; of wavefront 2: the rest
                    const int k = lane;
                    const int fi = p.imu_i[k], fj = p.imu_j[k];
                    double r15[15];
#pragma unroll
                    for (int q = 0; q < 15; ++q) r15[q] = 0.0;
                    if (!(p.fix[fi] == 3 && p.fix[fj] == 3)) {
                        const FState ci = load_state(CS + 16 * fi), cj = load_state(CS + 16 * fj);
                        const ImuRec cpre = load_imu(recs + k * XRHIP_IMU_DIM);
                        const V3 cbg = v3(bref[6 * k], bref[6 * k + 1], bref[6 * k + 2]);
                        const V3 cba = v3(bref[6 * k + 3], bref[6 * k + 4], bref[6 * k + 5]);
                        if (wave == 3) {
                            const V3 rq = imu_residual_rq(ci, cj, cpre, cbg, imu);
                            r15[0] = rq.x;
                            r15[1] = rq.y;
                            r15[2] = rq.z;
                        } else {
                            imu_residual_rest(ci, cj, cpre, cbg, cba, imu, r15);
                        }
                    }
                    if (wave == 3) {
                        for (int q = 0; q < 3; ++q) raw[15 * k + q] = r15[q];
                    } else {
                        for (int q = 3; q < 15; ++q) raw[15 * k + q] = r15[q];
                    }
                }
                __syncthreads();
                CPROF(15);  // trial: factor evaluations at the candidate
                for (int o = wtid; o < M; o += nt) red[0] += oc[o];
                for (int o = wtid; o < MR; o += nt) red[0] += oc[M + o];
                for (int it = wtid; it < NI * 15; it += nt) {
                    const int k = it / 15, i = it - 15 * k;
                    const double *S = recs + k * XRHIP_IMU_DIM + 56 + 15 * i;
                    double acc = 0;
#pragma unroll
                    for (int j = 0; j < 15; ++j) acc += S[j] * raw[15 * k + j];
                    red[0] += 0.5 * acc * acc;
                }
                for (int f = wtid; f < F; f += nt) {
                    const double *a = X + 16 * f, *b = CS + 16 * f;
                    double acc = 0;
                    if (pose_free(p.fix[f]))
                        for (int q = 0; q < 7; ++q) acc += (a[q] - b[q]) * (a[q] - b[q]);
                    if (motion_free(p.fix[f]))
                        for (int q = 7; q < 16; ++q) acc += (a[q] - b[q]) * (a[q] - b[q]);
                    red[1] += acc;
                }
                block_sum_n<2>(red, scratch);
                CPROF(16);  // trial: cost sums
                cost = red[0];
                dn2 = red[1];
                have_prev = true;
                prev_ca = ca;
                prev_cb = cb;
                prev_mcc = mcc;
                prev_cost = cost;
                prev_dn2 = dn2;
                prev_sn = step_norm;
            }
            if (trial_decide(t, 0, mcc, cost, dn2, step_norm)) accepted = 1;
        }
        if (accepted) {
            for (int e = wtid; e < 16 * F; e += nt) X[e] = CS[e];
            for (int e = wtid; e < 12 * nfree; e += nt) (&s_ftab[0][0][0])[e] = (&s_ftab[1][0][0])[e];   // the candidate's tables are x's now
        }
        if (tid == 0) trial_store(c, t);
        __syncthreads();
        CPROF(10);  // trials
#ifdef XRHIP_KPROF
        if (tid == 0) s_ctl.prof[27] += 1;   // rounds
#endif
        st = t.status;
        if (st == ST_DONE) break;
        if (st == ST_ACCEPTED) {
            relin = true;
            mode = 1;
        } else if (st == ST_RESOLVE || st == ST_RESOLVE_INNER) {
            relin = false;
            mode = (st == ST_RESOLVE) ? 2 : 3;
        } else {
            break;
        }
    }
    // -------------------- publication: states back to the arena, control block + states + sequence number to the host
    for (int e = wtid; e < 16 * F; e += nt) p.state[e] = X[e];
    __syncthreads();
#ifdef XRHIP_KPROF
    if (tid == 0) s_ctl.prof[12] += wall_clock64() - cp_t0;   // kernel entry -> publication
#endif
    // One pass: the optimised states and the control block go to the host mailbox straight from LDS, one word per thread
    // (the landmarks are constant in these problems: the host keeps its inverse depths), then the sequence number.  The
    // control block also returns to the arena (the next solve of this context reads radius / counters from there).
    const int fin = (st == ST_DONE) ? ST_DONE : -1;   // -1: not terminated within the round budget, the host reports an error
    if (tid == 0) c->status = fin;
    __syncthreads();
    for (int e = wtid; e < 16 * F; e += nt) p.host_out[e] = X[e];
    {
        constexpr unsigned ctl_words = sizeof(BaCtl) / sizeof(long long);
        const long long *src = reinterpret_cast<const long long *>(&s_ctl);
        long long *dst = reinterpret_cast<long long *>(static_cast<BaCtl *>(p.ctl));
        long long *hst = reinterpret_cast<long long *>(static_cast<BaCtl *>(p.host_ctl));
        for (unsigned i = tid; i < ctl_words; i += nt) {
            const long long v = src[i];
            dst[i] = v;
            hst[i] = v;
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) *reinterpret_cast<volatile int *>(static_cast<int *>(p.host_seq)) = seq;
}

}   // namespace xrhip
