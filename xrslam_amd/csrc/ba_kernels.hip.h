// ba_kernels.hip.h -- gfx950 kernels of the sliding-window visual-inertial bundle adjustment.
//
// Replaces what the reference delegates to Ceres 1.14 behind xrslam::Solver::solve()
// (/root/reference/xrslam/src/xrslam/estimation/solver.cpp:176-190): residual/Jacobian
// evaluation of the factor set, CauchyLoss correction, Jacobi scaling, Schur elimination of
// the inverse-depth blocks, the reduced-system Cholesky and the traditional dogleg loop.
//
// Launches of one trust-region round (all f64, everything stays in HBM/L2 between kernels):
//   kb_lin_all           per-factor residuals + Jacobians of all four factor families
//   kb_landmark_vision   per-landmark H_ll, g_l, dense cross-term row W_l; per-frame-pair reprojection blocks
//   kb_assemble          deterministic gather of all factor blocks into the (15F)^2 frame Hessian
//   kb_prepare           Jacobi scales, dogleg diagonal, landmark Schur weights
//   kb_schur_aux         T = W^T diag(omega) W on the f64 matrix cores + the solve's wide auxiliary passes; cost + gradient norm
//   kb_solve_try         reduced system, LDS-resident blocked Cholesky, dogleg data, trust-region trials
// The bodies are __device__ functions (..._item: one thread / wavefront per item, ..._block: one workgroup).
// Summation orders are fixed (no floating-point atomics), so results are run-to-run identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_math.hip.h"
#include "batch.hip.h"
#include "dense_lds.hip.h"

namespace xrhip {

constexpr int OREC = 28;   // per-observation record: Jt(12) Jr(12) jl(2) r(2), robustified
constexpr int RREC = 8;
constexpr int TRY_B = 4;   // trust-region trials costed per sweep after a rejection (inside kb_solve_try)
constexpr int WIDE_B = 8;  // ... per kb_trials_wide launch
constexpr int WIDE_G = 64; // workgroups of kb_trials_wide (128: the last block's 128 loads per sum and the ticket cost what the thinner slices save)
constexpr int QF_ROWS = 8; // landmark rows a wavefront keeps in flight in the back-substitution

struct BaCtl {   // device-resident solver state (one per context)
    double radius, mu;
    double x_cost, cand_cost, minimum_cost;
    double x_norm, gmax, alpha, step_norm;
    double model_cost_change, initial_cost;
    double q_gg, q_gn, q_nn;   // J-quadratic forms of the scaled gradient / Gauss-Newton directions (see kb_schur_aux)
    double gnorm, gn_norm, gd; // |grad|, |gn|, grad.gn of the current linearisation (kept for kb_trials_wide)
    int iteration, successful_steps, invalid_steps;
    int reuse;          // DoglegStrategy::reuse_
    int status;         // ST_* below
    int termination;    // XRHIP_BA_*
    int linear_ok;      // result of the last kb_solve
    int first;          // first linearisation of this solve (Jacobi scales are frozen afterwards)
    int max_iterations;
    unsigned wide_ticket;   // blocks of the running kb_trials_wide launch that have delivered their partial sums
    int accepted_slot;      // kb_trials_wide: candidate slot of the accepted step (-1: none); slot 0 of a first batch is the
                            // candidate spec_state_block linearised ahead of the decision
    long long prof[32];   // accumulated 100 MHz ticks per kernel phase (only written by -DXRHIP_KPROF builds)
};
// -DXRHIP_KPROF: phase sums in BaCtl::prof (tools/kprof_run.sh); -DXRHIP_KPROF_PRINT: per-workgroup timers of the multi-workgroup
// kernels, printed from the device (tools/gpu.sh kprint) -- how the load imbalances of kb_landmark_vision / kb_schur_aux /
// kb_trials_wide were found in round 3
#ifdef XRHIP_KPROF
#define KPROF_BEGIN() long long kp_t = wall_clock64()
#define KPROF(slot)                                                   \
    do {                                                              \
        __syncthreads();                                              \
        if (threadIdx.x == 0) {                                       \
            const long long kp_n = wall_clock64();                    \
            p.ctl->prof[slot] += kp_n - kp_t;                         \
            kp_t = kp_n;                                              \
        }                                                             \
    } while (0)
#else
#define KPROF_BEGIN() \
    do {              \
    } while (0)
#define KPROF(slot) \
    do {            \
    } while (0)
#endif
enum { ST_RUNNING = 0, ST_ACCEPTED = 1, ST_RESOLVE = 2, ST_DONE = 3, ST_NEED_TRIALS = 4, ST_RESOLVE_INNER = 102 };

struct BaDims {
    int F, n, PF;       // frames, 15F, pose dims padded to 16
    int L, Lp;          // landmarks, padded to 16
    int M, MR, NI;      // reprojection / rotation / imu factors
    int NP, np;         // prior frames, 15*NP
    int NV;             // n + L
    int robust;         // 1: CauchyLoss(1) on visual factors (Solver); 0: none (marginalisation)
    int na;             // number of free frame dofs (size of the reduced system actually factored)
    int nla;            // number of free landmarks (0: no Schur complement)
    int lm_rows;        // landmark rows kb_landmark_vision builds: Lp, or 0 when the solver has no free landmark
    int nfree;          // frames with a free pose or motion block
    int nffp;           // reprojection factors whose target AND reference pose are free (off-diagonal reprojection blocks)
    int schur_mode;     // 0 (always, in the product): f64 Schur contraction; 1 / 2: f32 / bf16 matrix-core operands -- BASELINE config 5's
                        // precision study only (xrhip_ba_debug_set_schur_precision)
    int sred_tiled;     // 1: the reduced system does not fit LDS and is factored where kb_schur_aux writes it, in the tiled layout of
                        // dense_lds.hip.h (round 6); 0: packed lower triangle (copied into LDS by the solve)
};

// Pointers of the argument blocks.  Kernels that receive BaPtrs by value see global-address-space pointers (the
// compiler knows that kernel arguments point to global memory); kb_tiny reads its argument block from device memory
// and would get generic pointers -- every access a FLAT instruction, which counts against vmcnt AND lgkmcnt, so a wait
// for a cross-lane shuffle also waits for all loads in flight (1217 flat_load/flat_store in kb_tiny, 0 global).
// The members are therefore wrappers around address-space-1 pointers, from which the compiler infers global loads /
// stores everywhere (A/B on MI355X, profiles/r02_ab_variants.md: localize_newframe 0.180 -> 0.173 ms per frame).
template <class T> struct gptr {
    // stored WITH its address space: a load of this member yields a global pointer, and the single cast to a generic
    // one below is what address-space inference starts from (a generic -> global -> generic cast pair on a plain
    // member is folded away before the inference runs, and an assumption "neither LDS nor scratch" was not picked up)
    __attribute__((address_space(1))) T *raw;
    __host__ __device__ __forceinline__ operator T *() const { return (T *)raw; }
    __host__ __device__ __forceinline__ T *operator->() const { return (T *)raw; }
    __host__ __device__ __forceinline__ gptr &operator=(T *p) {
        raw = (__attribute__((address_space(1))) T *)p;
        return *this;
    }
};

struct BaPtrs {
    // problem
    gptr<double> state, cand;            // [F][16], [TRY_B][F][16]
    gptr<const uint8_t> fix;              // [F]
    gptr<double> depth, depth_cand;      // [L], [TRY_B][L]
    gptr<const uint8_t> lact;             // [L] landmark is a free parameter
    gptr<const int> obs_tgt, obs_ref, obs_lm;
    gptr<const double> obs_zt, obs_zr;
    gptr<const int> rot_tgt, rot_ref;
    gptr<const double> rot_zt, rot_zr;
    gptr<const int> imu_i, imu_j;
    gptr<const double> imu_data;
    gptr<double> bias_ref;                // [NI][6]
    gptr<const int> prior_frames;
    gptr<const double> pS, pinfo, plin;
    gptr<double> pLam;                    // S^T S
    gptr<double> pc0;                     // S^T infovec   (both once per solve: kb_prior_lambda)
    // index structures (built on the host)
    gptr<const int> lm_start, lm_obs;    // CSR landmark -> observations
    gptr<const int> pair_start, pair_items;   // CSR (row frame, col frame) -> (obs << 1 | role)
    gptr<const int> rotf_start, rotf_items;   // CSR frame -> rotation factors
    gptr<const int> imuf;                 // [F][2]: imu factor with j == f, imu factor with i == f (or -1)
    gptr<const int> priorf;               // [F]: index in prior_frames or -1
    gptr<const int> act_idx;              // [na] free frame dofs (indices into [0, 15F)), ascending
    gptr<const int> act_inv;              // [15F] inverse of act_idx (-1: dof is constant)
    gptr<double> Hv, gv;                 // [F][F][36] reprojection blocks, [F][6] reprojection gradient
    // linearisation products
    gptr<double> orec, ocost;            // [M][28], [M]
    gptr<double> rrec, rcost;
    gptr<double> imu_r, imu_Ji, imu_Jj, imu_cost;   // [NI][15], [NI][225] x2, [NI]
    gptr<double> pr, pt, pJq, pcost;   // prior residual [np], S^T r [np], Jr^-1 [NP][9], cost [1 + ceil(np / 16)]: total | per row block
    gptr<double> Hpp, gp;                // [n][n], [n]   unscaled frame Hessian / gradient
    gptr<double> hll, gl, Wt;           // [L], [L], [Lp][PF]
    gptr<double> sp, sl, omega;         // Jacobi scales [n], [L]; Schur weights [L]
    gptr<double> T;                       // [PF][PF]
    gptr<double> Sred;                    // [n][n] scratch (global fallback of the Cholesky)
    gptr<double> diagD, grad, gn, gs, step, delta;   // [NV] each
    gptr<double> partial;                 // [aux_quad_blocks] partial sums of Q(g~,g~)
    gptr<double> wog;                     // [WOG_CH][PF] W^T (omega gl), per landmark-row chunk (wog_at)
    gptr<double> wide_part;               // [WIDE_G][4 WIDE_B] per-block partial sums of kb_trials_wide
    // zero-copy mailbox in pinned host memory (device-visible addresses): the trial kernel publishes the
    // control block, on termination the optimised states, and last a sequence number the host spins on
    gptr<BaCtl> host_ctl;
    gptr<double> host_out;                // [16 F + L]
    gptr<int> host_seq;
    gptr<BaCtl> ctl;
};

XD bool pose_free(uint8_t fix) { return !(fix & 1); }
XD bool motion_free(uint8_t fix) { return !(fix & 2); }
XD bool dof_active(const uint8_t *fix, int a) {   // a in [0, 15F)
    const int f = a / 15, k = a - 15 * f;
    return k < 6 ? pose_free(fix[f]) : motion_free(fix[f]);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// deterministic block-wide sum (fixed tree); scratch must hold blockDim.x/64 doubles
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; ++i) s += scratch[i];
    __syncthreads();
    return s;
}

// deterministic block-wide sum of N values at once (2 barriers); scratch must hold N * blockDim.x/64 doubles
template <int N> __device__ __forceinline__ void block_sum_n(double (&v)[N], double *scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < N; ++k) scratch[k * nw + w] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = 0;
        for (int i = 0; i < nw; ++i) s += scratch[k * nw + i];
        v[k] = s;
    }
}

// --------------------------------------------------------------- reprojection factors
// One thread per observation.  use_cand selects the candidate state (cost only).
// cost only, with the landmark's inverse depth given by value (kb_trials_wide forms candidates on the fly)
__device__ __forceinline__ double obs_cost_at(const BaDims &d, const BaPtrs &p, int o, const double *state, double inv_depth,
                                              const Ext &cam, double sx, double sy) {
    const int ft = p.obs_tgt[o], fr = p.obs_ref[o], l = p.obs_lm[o];
    if (!pose_free(p.fix[ft]) && !pose_free(p.fix[fr]) && !p.lact[l]) return 0.0;
    const FState st = load_state(state + 16 * ft), sr = load_state(state + 16 * fr);
    const V3 zt = v3(p.obs_zt[3 * o], p.obs_zt[3 * o + 1], p.obs_zt[3 * o + 2]);
    const V3 zr = v3(p.obs_zr[3 * o], p.obs_zr[3 * o + 1], p.obs_zr[3 * o + 2]);
    double r[2];
    eval_reprojection(st, sr, inv_depth, zt, zr, cam, sx, sy, r, false, nullptr, nullptr, nullptr);
    const double s2 = r[0] * r[0] + r[1] * r[1];
    return d.robust ? 0.5 * log(1.0 + s2) : 0.5 * s2;
}

__device__ __forceinline__ double obs_eval(const BaDims &d, const BaPtrs &p, int o, const double *state,
                                           const double *depth, const Ext &cam, double sx, double sy, bool want_j,
                                           double *rec) {
    const int ft = p.obs_tgt[o], fr = p.obs_ref[o], l = p.obs_lm[o];
    const bool at = pose_free(p.fix[ft]), ar = pose_free(p.fix[fr]), al = p.lact[l] != 0;
    if (!at && !ar && !al) {   // constant residual block: removed by Ceres' preprocessor
        if (want_j)
            for (int i = 0; i < OREC; ++i) rec[i] = 0.0;
        return 0.0;
    }
    const FState st = load_state(state + 16 * ft), sr = load_state(state + 16 * fr);
    const V3 zt = v3(p.obs_zt[3 * o], p.obs_zt[3 * o + 1], p.obs_zt[3 * o + 2]);
    const V3 zr = v3(p.obs_zr[3 * o], p.obs_zr[3 * o + 1], p.obs_zr[3 * o + 2]);
    double r[2], Jt[12], Jr[12], Jl[2];
    eval_reprojection(st, sr, depth[l], zt, zr, cam, sx, sy, r, want_j, Jt, Jr, Jl);
    const double s = r[0] * r[0] + r[1] * r[1];
    if (want_j) {
        // CauchyLoss(1): rho' = 1/(1+s), rho'' < 0  =>  residual and Jacobian scaled by sqrt(rho')
        const double sc = d.robust ? sqrt(fmax(2.2250738585072014e-308, 1.0 / (1.0 + s))) : 1.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            rec[i] = at ? Jt[i] * sc : 0.0;
            rec[12 + i] = ar ? Jr[i] * sc : 0.0;
        }
        rec[24] = al ? Jl[0] * sc : 0.0;
        rec[25] = al ? Jl[1] * sc : 0.0;
        rec[26] = r[0] * sc;
        rec[27] = r[1] * sc;
    }
    return d.robust ? 0.5 * log(1.0 + s) : 0.5 * s;
}

__device__ __forceinline__ void lin_obs_item(const BaDims &d, const BaPtrs &p, const Ext &cam, double sx, double sy, int o) {
    double rec[OREC];
    p.ocost[o] = obs_eval(d, p, o, p.state, p.depth, cam, sx, sy, true, rec);
#pragma unroll
    for (int i = 0; i < OREC; ++i) p.orec[(size_t)o * OREC + i] = rec[i];
}

__device__ __forceinline__ double rot_eval(const BaDims &d, const BaPtrs &p, int o, const double *state,
                                           const Ext &cam, double sx, double sy, bool want_j, double *rec) {
    const int ft = p.rot_tgt[o], fr = p.rot_ref[o];
    if (!pose_free(p.fix[ft])) {
        if (want_j)
            for (int i = 0; i < RREC; ++i) rec[i] = 0.0;
        return 0.0;
    }
    const FState st = load_state(state + 16 * ft), sr = load_state(state + 16 * fr);
    double r[2], Jq[6];
    eval_rotation(st, sr, v3(p.rot_zt[3 * o], p.rot_zt[3 * o + 1], p.rot_zt[3 * o + 2]),
                  v3(p.rot_zr[3 * o], p.rot_zr[3 * o + 1], p.rot_zr[3 * o + 2]), cam, sx, sy, r, want_j, Jq);
    const double s = r[0] * r[0] + r[1] * r[1];
    if (want_j) {
        const double sc = sqrt(fmax(2.2250738585072014e-308, 1.0 / (1.0 + s)));
#pragma unroll
        for (int i = 0; i < 6; ++i) rec[i] = Jq[i] * sc;
        rec[6] = r[0] * sc;
        rec[7] = r[1] * sc;
    }
    return 0.5 * log(1.0 + s);
}

__device__ __forceinline__ void lin_rot_item(const BaDims &d, const BaPtrs &p, const Ext &cam, double sx, double sy, int o) {
    double rec[RREC];
    p.rcost[o] = rot_eval(d, p, o, p.state, cam, sx, sy, true, rec);
#pragma unroll
    for (int i = 0; i < RREC; ++i) p.rrec[(size_t)o * RREC + i] = rec[i];
}

// --------------------------------------------------------------------- IMU factors
// whitened residual only (15 values via out) -- used for costs; one thread does everything
__device__ __forceinline__ double imu_cost_eval(const BaPtrs &p, int k, const double *state, const Ext &imu) {
    const int fi = p.imu_i[k], fj = p.imu_j[k];
    if (p.fix[fi] == 3 && p.fix[fj] == 3) return 0.0;
    const double *data = p.imu_data + (size_t)k * XRHIP_IMU_DIM;
    const ImuRec pre = load_imu(data);
    double raw[15];
    imu_raw_residual(load_state(state + 16 * fi), load_state(state + 16 * fj), pre,
                     v3(p.bias_ref[6 * k], p.bias_ref[6 * k + 1], p.bias_ref[6 * k + 2]),
                     v3(p.bias_ref[6 * k + 3], p.bias_ref[6 * k + 4], p.bias_ref[6 * k + 5]), imu, raw);
    const double *S = data + 56;
    double c = 0;
    for (int i = 0; i < 15; ++i) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < 15; ++j) s += S[15 * i + j] * raw[j];
        c += s * s;
    }
    return 0.5 * c;
}

// wave-cooperative whitened IMU cost: returns the factor's cost in lane 0 (0 in the other lanes)
__device__ __forceinline__ double imu_cost_wave(const BaPtrs &p, int k, const double *state, const Ext &imu, int lane) {
    const int fi = p.imu_i[k], fj = p.imu_j[k];
    if (p.fix[fi] == 3 && p.fix[fj] == 3) return 0.0;
    const double *data = p.imu_data + (size_t)k * XRHIP_IMU_DIM;
    double raw[15];
    if (lane == 0) {
        const ImuRec pre = load_imu(data);
        imu_raw_residual(load_state(state + 16 * fi), load_state(state + 16 * fj), pre,
                         v3(p.bias_ref[6 * k], p.bias_ref[6 * k + 1], p.bias_ref[6 * k + 2]),
                         v3(p.bias_ref[6 * k + 3], p.bias_ref[6 * k + 4], p.bias_ref[6 * k + 5]), imu, raw);
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) raw[i] = __shfl(raw[i], 0);
    double c = 0;
    if (lane < 15) {
        const double *S = data + 56 + 15 * lane;
        double s = 0;
#pragma unroll
        for (int j = 0; j < 15; ++j) s += S[j] * raw[j];
        c = 0.5 * s * s;
    }
    return wave_sum(c) * (lane == 0 ? 1.0 : 0.0);
}

// One wavefront per IMU factor: lane 0 evaluates the (serial) SO(3) algebra into LDS, then all lanes apply the
// 15x15 whitening to the residual and both Jacobians.  scr: IMU_SCR doubles of LDS owned by this wavefront.
constexpr int IMU_SCR = 15 + 225 + 225;
__device__ __forceinline__ void lin_imu_item(const BaDims &d, const BaPtrs &p, const Ext &imu, int k, int lane, double *scr) {
    double *raw = scr, *Ji = scr + 15, *Jj = scr + 240;
    const int fi = p.imu_i[k], fj = p.imu_j[k];
    const double *data = p.imu_data + (size_t)k * XRHIP_IMU_DIM;
    const bool active = !(p.fix[fi] == 3 && p.fix[fj] == 3);
    for (int i = lane; i < 225; i += 64) {
        Ji[i] = 0.0;
        Jj[i] = 0.0;
    }
    wave_sync();
    if (lane == 0 && active) {
        const FState si = load_state(p.state + 16 * fi), sj = load_state(p.state + 16 * fj);
        const ImuRec pre = load_imu(data);
        const V3 bg0 = v3(p.bias_ref[6 * k], p.bias_ref[6 * k + 1], p.bias_ref[6 * k + 2]);
        const V3 ba0 = v3(p.bias_ref[6 * k + 3], p.bias_ref[6 * k + 4], p.bias_ref[6 * k + 5]);
        double r15[15];
        imu_raw_residual(si, sj, pre, bg0, ba0, imu, r15);
        for (int i = 0; i < 15; ++i) raw[i] = r15[i];
        imu_raw_jacobians(si, sj, pre, bg0, ba0, imu, v3(r15[0], r15[1], r15[2]), Ji, Jj, p.fix[fi] != 3, p.fix[fj] != 3);
    }
    wave_sync();
    const double *S = data + 56;
    double cost = 0.0;
    if (lane < 15) {
        double s = 0;
        if (active)
            for (int j = 0; j < 15; ++j) s += S[15 * lane + j] * raw[j];
        p.imu_r[15 * k + lane] = s;
        cost = 0.5 * s * s;
    }
    cost = wave_sum(cost);
    if (lane == 0) p.imu_cost[k] = cost;
    for (int e = lane; e < 225; e += 64) {
        const int i = e / 15, c = e - 15 * i;
        double a = 0, b = 0;
        if (active) {
            // constant blocks contribute no columns
            const bool col_i = c < 6 ? pose_free(p.fix[fi]) : motion_free(p.fix[fi]);
            const bool col_j = c < 6 ? pose_free(p.fix[fj]) : motion_free(p.fix[fj]);
            if (col_i)
                for (int j = 0; j < 15; ++j) a += S[15 * i + j] * Ji[15 * j + c];
            if (col_j)
                for (int j = 0; j < 15; ++j) b += S[15 * i + j] * Jj[15 * j + c];
        }
        p.imu_Ji[(size_t)225 * k + e] = a;
        p.imu_Jj[(size_t)225 * k + e] = b;
    }
    wave_sync();   // scr may be reused by the caller for the next factor
}

// One WORKGROUP (four wavefronts) per IMU factor: lane 0 of every wavefront runs the raw residual (they all need its
// rotation part), then one of the four independent parts of the raw Jacobians; the 15x15 whitening of the residual and of
// both Jacobians is then one pass over the 256 threads.  Same expressions, entry by entry, as lin_imu_item -- which gave
// a factor ONE wavefront, i.e. one lane for ~25 us of dependent SO(3) algebra, the long pole of kb_lin_all.
// scr: IMU_SCR + IMU_XCH doubles of LDS owned by the workgroup.
__device__ __forceinline__ void lin_imu_block(const BaDims &d, const BaPtrs &p, const Ext &imu, int k, double *scr) {
    double *raw = scr, *Ji = scr + 15, *Jj = scr + 240;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = p.imu_i[k], fj = p.imu_j[k];
    const double *data = p.imu_data + (size_t)k * XRHIP_IMU_DIM;
    const bool active = !(p.fix[fi] == 3 && p.fix[fj] == 3);
    for (int i = tid; i < 450; i += 256) Ji[i] = 0.0;   // Ji and Jj are contiguous
    __syncthreads();
    // kb_chain's schedule (ba_chain.hip.h; the pieces are bit-identical to the single-lane forms, tests/test_ba_math_host.py): the
    // rotation residual -- the long chain -- runs ONCE, on wavefront 3, followed by what needs only it (Jr^-1, B) and the rest of the
    // residual; the other wavefronts meanwhile form the parts of the Jacobians that do not depend on it; two short products finish.
    // (Before: every wavefront's lane 0 ran the whole residual, then one of four Jacobian parts: ~8 of kb_lin_all's 11 us.)
    double *xch = scr + IMU_SCR;
    const bool nfi = p.fix[fi] != 3, nfj = p.fix[fj] != 3;
    if (lane == 0 && active) {
        const FState si = load_state(p.state + 16 * fi), sj = load_state(p.state + 16 * fj);
        const ImuRec pre = load_imu(data);
        const V3 bg0 = v3(p.bias_ref[6 * k], p.bias_ref[6 * k + 1], p.bias_ref[6 * k + 2]);
        const V3 ba0 = v3(p.bias_ref[6 * k + 3], p.bias_ref[6 * k + 4], p.bias_ref[6 * k + 5]);
        if (wave == 3) {
            const V3 rq = imu_residual_rq(si, sj, pre, bg0, imu);
            raw[0] = rq.x;
            raw[1] = rq.y;
            raw[2] = rq.z;
            store33(xch + 45, imu_jac_jrinv(rq));
            if (nfi) store33(xch + 36, imu_jac_B(rq));
            imu_residual_rest(si, sj, pre, bg0, ba0, imu, raw);
        } else if (wave == 0) {
            imu_raw_jacobians_part(2, si, sj, pre, bg0, ba0, imu, v3(0, 0, 0), Ji, Jj, nfi, nfj);
        } else if (wave == 1) {
            imu_jac_pre(si, sj, pre, bg0, imu, xch, nfi, nfj);
        } else {
            imu_raw_jacobians_part(3, si, sj, pre, bg0, ba0, imu, v3(0, 0, 0), Ji, Jj, nfi, nfj);
        }
    }
    __syncthreads();
    if (lane == 0 && active) {
        if (wave == 0) imu_jac_finish0(load33(xch + 45), xch, Ji, Jj, nfi, nfj);
        else if (wave == 1) imu_jac_finish1(load33(xch + 45), load33(xch + 36), xch, load33(data + 11), Ji, nfi);
    }
    __syncthreads();
    const double *S = data + 56;
    if (wave == 0) {
        double cost = 0.0;
        if (lane < 15) {
            double s = 0;
            if (active)
                for (int j = 0; j < 15; ++j) s += S[15 * lane + j] * raw[j];
            p.imu_r[15 * k + lane] = s;
            cost = 0.5 * s * s;
        }
        cost = wave_sum(cost);
        if (lane == 0) p.imu_cost[k] = cost;
    }
    if (tid < 225) {
        const int e = tid, i = e / 15, c = e - 15 * i;
        double a = 0, b = 0;
        if (active) {
            // constant blocks contribute no columns
            const bool col_i = c < 6 ? pose_free(p.fix[fi]) : motion_free(p.fix[fi]);
            const bool col_j = c < 6 ? pose_free(p.fix[fj]) : motion_free(p.fix[fj]);
            if (col_i)
                for (int j = 0; j < 15; ++j) a += S[15 * i + j] * Ji[15 * j + c];
            if (col_j)
                for (int j = 0; j < 15; ++j) b += S[15 * i + j] * Jj[15 * j + c];
        }
        p.imu_Ji[(size_t)225 * k + e] = a;
        p.imu_Jj[(size_t)225 * k + e] = b;
    }
}

// ------------------------------------------------------------------------ prior
// delta and Jr^-1 of prior frame i at `state`
__device__ __forceinline__ void prior_delta(const BaPtrs &p, int i, const double *state, double *delta15, M3 *Jq) {
    const FState s = load_state(state + 16 * p.prior_frames[i]);
    const FState l = load_state(p.plin + 16 * i);
    const V3 rq = logmap(q_mul(q_conj(l.q), s.q));
    delta15[0] = rq.x; delta15[1] = rq.y; delta15[2] = rq.z;
    delta15[3] = s.p.x - l.p.x; delta15[4] = s.p.y - l.p.y; delta15[5] = s.p.z - l.p.z;
    delta15[6] = s.v.x - l.v.x; delta15[7] = s.v.y - l.v.y; delta15[8] = s.v.z - l.v.z;
    delta15[9] = s.bg.x - l.bg.x; delta15[10] = s.bg.y - l.bg.y; delta15[11] = s.bg.z - l.bg.z;
    delta15[12] = s.ba.x - l.ba.x; delta15[13] = s.ba.y - l.ba.y; delta15[14] = s.ba.z - l.ba.z;
    if (Jq) *Jq = inverse3(right_jacobian(rq));
}

// prior cost at `state`: 0.5 |S delta + infovec|^2 ; block-wide (any block size), delta staged in `sh` (np doubles).
// One wavefront per row of S, lanes striding over the columns (coalesced).
__device__ __forceinline__ double prior_cost_block(const BaDims &d, const BaPtrs &p, const double *state, double *sh,
                                                   double *scratch, double *r_out) {
    for (int i = threadIdx.x; i < d.NP; i += blockDim.x) {
        double dl[15];
        prior_delta(p, i, state, dl, nullptr);
        for (int k = 0; k < 15; ++k) sh[15 * i + k] = dl[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double c = 0;
    for (int i0 = 4 * wave; i0 < d.np; i0 += 4 * nw) {   // four rows per wavefront in flight
        double s4[4] = {0, 0, 0, 0};
        for (int j = lane; j < d.np; j += 64) {
            const double x = sh[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) s4[r] += p.pS[(size_t)min(i0 + r, d.np - 1) * d.np + j] * x;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s4[r] = wave_sum(s4[r]);
        if (lane == 0)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i0 + r < d.np) {
                    const double t = s4[r] + p.pinfo[i0 + r];
                    if (r_out) r_out[i0 + r] = t;
                    c += t * t;
                }
    }
    return 0.5 * block_sum(c, scratch);
}

// The same linearisation spread over ceil(np / 16) workgroups of 256 (kb_lin_all): the single workgroup above walks the 150 x 150
// matrix twice with four rows per wavefront in flight and a butterfly per row -- 25 us, the longest block of kb_lin_all by far.
// Block b owns rows 16 b .. 16 b + 15 of r = S delta + infovec AND of t = S^T r, the latter from the per-solve products
// Lam = S^T S and c0 = S^T infovec (kb_prior_lambda):  t = Lam delta + c0.  Eight threads per row and matrix (32 dot products of
// length np per block), three butterfly stages.  Every block recomputes delta (a logmap per prior frame); block 0 also stores the
// Jr^-1 blocks.  The cost is left as per-block partial sums pcost[1 + b] (sum_cost_block adds them in block order).
__host__ __device__ __forceinline__ int prior_row_blocks(int np) { return (np + 15) / 16; }
__device__ __forceinline__ void lin_prior_rows_block(const BaDims &d, const BaPtrs &p, int b, double *sh, double *scratch) {
    const int tid = threadIdx.x, np = d.np;
    for (int i = tid; i < d.NP; i += blockDim.x) {
        double dl[15];
        M3 Jq;
        prior_delta(p, i, p.state, dl, &Jq);
        for (int k = 0; k < 15; ++k) sh[15 * i + k] = dl[k];
        if (b == 0)
            for (int k = 0; k < 9; ++k) p.pJq[9 * i + k] = Jq.m[k];
    }
    __syncthreads();
    const int dot = tid >> 3, part = tid & 7;          // 32 dot products: 0-15 rows of S, 16-31 rows of Lam
    const int row = 16 * b + (dot & 15);
    const double *M = (dot < 16 ? static_cast<const double *>(p.pS) : static_cast<const double *>(p.pLam)) + (size_t)min(row, np - 1) * np;
    double s = 0.0;
    for (int j0 = part; j0 < np; j0 += 64) {            // eight loads in flight
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = M[min(j0 + 8 * u, np - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + 8 * u < np) s += v[u] * sh[j0 + 8 * u];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    double c = 0.0;
    if (part == 0 && row < np) {
        if (dot < 16) {
            const double r = s + p.pinfo[row];
            p.pr[row] = r;
            c = r * r;
        } else {
            p.pt[row] = s + p.pc0[row];
        }
    }
    c = block_sum(c, scratch);
    if (tid == 0) {
        p.pcost[1 + b] = 0.5 * c;
        if (b == 0) p.pcost[0] = 0.0;
    }
}

// block-wide, any workgroup size; sh: np doubles of LDS, scratch: blockDim/64 doubles
__device__ __forceinline__ void lin_prior_block(const BaDims &d, const BaPtrs &p, double *sh, double *scratch) {
    if (d.NP == 0) {
        if (threadIdx.x == 0) p.pcost[0] = 0.0;
        return;
    }
    for (int i = threadIdx.x; i < d.NP; i += blockDim.x) {
        double dl[15];
        M3 Jq;
        prior_delta(p, i, p.state, dl, &Jq);
        for (int k = 0; k < 9; ++k) p.pJq[9 * i + k] = Jq.m[k];
    }
    const double c = prior_cost_block(d, p, p.state, sh, scratch, p.pr);
    if (threadIdx.x == 0) p.pcost[0] = c;
    for (int b = threadIdx.x; b < prior_row_blocks(d.np); b += blockDim.x) p.pcost[1 + b] = 0.0;   // (the row-block form's partial sums)
    __syncthreads();
    // t = S^T r  (thread per column: consecutive threads read consecutive addresses; eight rows per round trip)
    for (int j = threadIdx.x; j < d.np; j += blockDim.x) {
        double s = 0;
        for (int i0 = 0; i0 < d.np; i0 += 8) {
            double v[8], r8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = min(i0 + q, d.np - 1);
                v[q] = p.pS[(size_t)i * d.np + j];
                r8[q] = (i0 + q < d.np) ? p.pr[i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q] * r8[q];
        }
        p.pt[j] = s;
    }
}

// Lam = S^T S (once per solve with a prior) on the f64 matrix cores: one workgroup per 16x16 tile of Lam, the four
// wavefronts split the contraction (rows of S) and combine in LDS in a fixed order.
__global__ __launch_bounds__(256) void kb_prior_lambda(int np, const double *__restrict__ S, double *__restrict__ Lam,
                                                       const double *__restrict__ info, double *__restrict__ c0) {
    __shared__ double red[4][256];
    const int tiles = (np + 15) / 16;
    if ((int)blockIdx.x >= tiles * tiles) {   // `tiles` extra blocks: c0 = S^T infovec, sixteen columns each -- thread (column jl, row part):
        const int jl = threadIdx.x & 15, part = threadIdx.x >> 4, j = 16 * ((int)blockIdx.x - tiles * tiles) + jl;   // rows part, part + 16, ...
        double s = 0;
        for (int i = part; i < np; i += 16) s += S[(size_t)i * np + min(j, np - 1)] * info[i];
        red[0][threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 16 && j < np) {
            double t = 0;
            for (int q = 0; q < 16; ++q) t += red[0][16 * q + jl];
            c0[j] = t;
        }
        return;
    }
    const int ti = blockIdx.x / tiles, tj = blockIdx.x - ti * tiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int kq = (((np + 3) / 4 + 3) / 4) * 4;   // contraction rows per wavefront, multiple of 4
    const int ca = 16 * ti + i, cb = 16 * tj + i;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k = wave * kq; k < (wave + 1) * kq; k += 4) {
        const int r = k + kk;
        const double a = (r < np && ca < np) ? S[(size_t)r * np + ca] : 0.0;
        const double b = (r < np && cb < np) ? S[(size_t)r * np + cb] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(kk + 4 * r) * 16 + i] = acc[r];
    __syncthreads();
    const int e = threadIdx.x;
    const double v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    const int row = 16 * ti + (e >> 4), col = 16 * tj + (e & 15);
    if (row < np && col < np) Lam[(size_t)row * np + col] = v;
}

// -------------------------------------------------------------------- landmarks
// One wavefront per landmark: H_ll, g_l and the cross-term row W_l (6 values per observing frame),
// written as a dense row of Wt [Lp][PF] (zero elsewhere) so the Schur product is a plain MFMA SYRK.
__device__ __forceinline__ void landmark_item(const BaDims &d, const BaPtrs &p, int l, int lane) {
    double *row = p.Wt + (size_t)l * d.PF;
    for (int c = lane; c < d.PF; c += 64) row[c] = 0.0;
    if (l >= d.L) return;
    const int b = p.lm_start[l], e = p.lm_start[l + 1];
    double hll = 0, gl = 0, wr[6] = {0, 0, 0, 0, 0, 0};
    int ref = -1;
    wave_sync();
    for (int it = b + lane; it < e + ((64 - (e - b) % 64) % 64); it += 64) {   // uniform trip count for the shuffles
        const bool valid = it < e;
        double wt[6] = {0, 0, 0, 0, 0, 0};
        int ft = -1;
        if (valid) {
            const int o = p.lm_obs[it];
            const double *rec = p.orec + (size_t)o * OREC;
            const double j0 = rec[24], j1 = rec[25];
            hll += j0 * j0 + j1 * j1;
            gl += j0 * rec[26] + j1 * rec[27];
            ft = p.obs_tgt[o];
            ref = p.obs_ref[o];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                wt[a] = rec[a] * j0 + rec[6 + a] * j1;
                wr[a] += rec[12 + a] * j0 + rec[18 + a] * j1;
            }
        }
        // a frame observes a landmark at most once, so target rows never collide
        if (valid && p.lact[l])
#pragma unroll
            for (int a = 0; a < 6; ++a) row[6 * ft + a] = wt[a];
    }
    hll = wave_sum(hll);
    gl = wave_sum(gl);
#pragma unroll
    for (int a = 0; a < 6; ++a) wr[a] = wave_sum(wr[a]);
    int refm = ref;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) refm = max(refm, __shfl_xor(refm, off));
    __builtin_amdgcn_s_waitcnt(0);   // the row stores above must have landed before lane 0 adds the reference block
    wave_sync();
    if (lane == 0) {
        p.hll[l] = hll;
        p.gl[l] = gl;
        if (refm >= 0 && p.lact[l])
#pragma unroll
            for (int a = 0; a < 6; ++a) row[6 * refm + a] += wr[a];
    }
}

// --------------------------------------------------------------------- assembly
// Reprojection blocks: one wavefront per (row frame, column frame) pair.  The lanes stride over the pair's
// observation list, each accumulating a private 6x6 block (+ 6-vector for the diagonal pair); a fixed
// butterfly reduction combines them -- "batched small-block JtJ accumulation with wavefront-shuffle reductions".
constexpr int VIS_RED = 42 * 65;   // doubles of LDS assemble_vision_item needs per wavefront
// A pair's observation list is cut into VIS_CH chunks of >= 64 entries, one wavefront (workgroup of kb_landmark_vision) each; the
// chunks' partial blocks are added by the reader in chunk order (vis_h / vis_g).  As ONE wavefront per pair the diagonal pair of the
// oldest keyframe -- the reference frame of most landmarks, > 1000 entries -- ran 17-21 us while every other workgroup of the launch
// was done within 7 (in-kernel block timers, round 3): it alone set kb_landmark_vision's 22.7 us.
constexpr int VIS_CH = 8;
__device__ __forceinline__ double vis_h(const BaDims &d, const BaPtrs &p, int pair, int e) {
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < VIS_CH; ++ch) s += p.Hv[((size_t)ch * d.F * d.F + pair) * 36 + e];
    return s;
}
__device__ __forceinline__ double vis_g(const BaDims &d, const BaPtrs &p, int a) {
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < VIS_CH; ++ch) s += p.gv[(size_t)ch * 6 * d.F + a];
    return s;
}
// ch / nch: this wavefront's chunk of the pair's list (nch == 1: the whole list into chunk 0, the other chunks cleared)
__device__ __forceinline__ void assemble_vision_item(const BaDims &d, const BaPtrs &p, int pair, int lane, double *red, int ch = 0, int nch = 1) {
    const int fa = pair / d.F, fb = pair - fa * d.F;
    if (!pose_free(p.fix[fa]) || !pose_free(p.fix[fb])) return;   // the block is never read (kb_assemble)
    const size_t FF = (size_t)d.F * d.F;
    int s0 = p.pair_start[fa * d.F + fb], s1 = p.pair_start[fa * d.F + fb + 1];
    if (nch > 1) {
        const int per = max(64, (s1 - s0 + nch - 1) / nch);
        s0 = min(s1, s0 + ch * per);
        s1 = min(s1, s0 + per);
    } else {
        for (int c = 1; c < VIS_CH; ++c) {
            if (lane < 36) p.Hv[(c * FF + pair) * 36 + lane] = 0.0;
            if (fa == fb && lane < 6) p.gv[(size_t)c * 6 * d.F + 6 * fa + lane] = 0.0;
        }
    }
    if (s0 == s1) {   // no shared observation (always the case off the diagonal when the landmarks' reference frames are constant)
        if (lane < 36) p.Hv[(ch * FF + pair) * 36 + lane] = 0.0;
        if (fa == fb && lane < 6) p.gv[(size_t)ch * 6 * d.F + 6 * fa + lane] = 0.0;
        return;
    }
    double h[36], g[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) h[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
    const bool diag = (fa == fb);
    for (int it = s0 + lane; it < s1; it += 64) {
        const int code = p.pair_items[it];
        const double *rec = p.orec + (size_t)(code >> 1) * OREC;
        const int ra = (code & 1) ? 12 : 0;
        const int rb = diag ? ra : ((code & 1) ? 0 : 12);
        double ja[12], jb[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            ja[i] = rec[ra + i];
            jb[i] = rec[rb + i];
        }
        const double r0 = rec[26], r1 = rec[27];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b) h[6 * a + b] += ja[a] * jb[b] + ja[6 + a] * jb[6 + b];
            if (diag) g[a] += ja[a] * r0 + ja[6 + a] * r1;
        }
    }
    // The lanes' private blocks are combined through LDS: every lane parks its 42 partial sums as a column, lane e then
    // adds up row e over the lanes that had observations, in lane order.  (The first version ran 42 butterfly reductions
    // of doubles per pair -- 12 ds_bpermute each -- which was most of kb_landmark_vision's 25 us on a 10-keyframe window.)
    wave_sync();   // a previous pair's rows have been read
#pragma unroll
    for (int i = 0; i < 36; ++i) red[i * 65 + lane] = h[i];
    if (diag) {
#pragma unroll
        for (int i = 0; i < 6; ++i) red[(36 + i) * 65 + lane] = g[i];
    }
    wave_sync();
    const int nl = min(64, s1 - s0);
    if (lane < (diag ? 42 : 36)) {
        const double *row = red + lane * 65;
        double acc = 0.0;
        for (int q = 0; q < nl; ++q) acc += row[q];
        if (lane < 36) p.Hv[(ch * FF + pair) * 36 + lane] = acc;
        else p.gv[(size_t)ch * 6 * d.F + 6 * fa + lane - 36] = acc;
    }
}

// One thread per element (a,b) of the frame Hessian Hpp (15F x 15F) and, for b == 0, of g: adds the
// reprojection block, rotation priors, the (at most two) IMU factors adjacent to the frame and the prior,
// always in this fixed order.
__device__ __forceinline__ void assemble_item(const BaDims &d, const BaPtrs &p, int e) {
    const int a = e / d.n, b = e - a * d.n;
    const int fa = a / 15, ka = a - 15 * fa, fb = b / 15, kb = b - 15 * fb;
    double h = 0.0, g = 0.0;
    const bool want_g = (b == 0);
    if (dof_active(p.fix, a) && (dof_active(p.fix, b) || want_g)) {
        const bool bact = dof_active(p.fix, b);
        if (ka < 6) {
            if (kb < 6 && bact) h += vis_h(d, p, fa * d.F + fb, 6 * ka + kb);
            if (want_g) g += vis_g(d, p, 6 * fa + ka);
            if (ka < 3) {
                const int s = p.rotf_start[fa], t = p.rotf_start[fa + 1];
                for (int it = s; it < t; ++it) {
                    const double *rec = p.rrec + (size_t)p.rotf_items[it] * RREC;
                    if (fb == fa && kb < 3 && bact) h += rec[ka] * rec[kb] + rec[3 + ka] * rec[3 + kb];
                    if (want_g) g += rec[ka] * rec[6] + rec[3 + ka] * rec[7];
                }
            }
        }
        // IMU factors: side 0 has j == fa, side 1 has i == fa
        for (int side = 0; side < 2; ++side) {
            const int k = p.imuf[2 * fa + side];
            if (k < 0) continue;
            const double *Ja = (side == 0 ? p.imu_Jj : p.imu_Ji) + (size_t)225 * k;
            const int fi = p.imu_i[k], fj = p.imu_j[k];
            if (want_g) {
                const double *r = p.imu_r + 15 * k;
                for (int i = 0; i < 15; ++i) g += Ja[15 * i + ka] * r[i];
            }
            if (bact && (fb == fi || fb == fj)) {
                const double *Jb = (fb == fj ? p.imu_Jj : p.imu_Ji) + (size_t)225 * k;
                for (int i = 0; i < 15; ++i) h += Ja[15 * i + ka] * Jb[15 * i + kb];
            }
        }
        // prior: H = B^T Lam B, g = B^T S^T r with B = blockdiag(Jr^-1 on q rows)
        const int pa = p.priorf[fa];
        if (pa >= 0) {
            if (want_g) {
                if (ka < 3) {
                    for (int k = 0; k < 3; ++k) g += p.pJq[9 * pa + 3 * k + ka] * p.pt[15 * pa + k];
                } else {
                    g += p.pt[15 * pa + ka];
                }
            }
            const int pb = p.priorf[fb];
            if (pb >= 0 && bact) {
                double v = 0;
                if (ka < 3 && kb < 3) {
                    for (int k = 0; k < 3; ++k)
                        for (int m = 0; m < 3; ++m)
                            v += p.pJq[9 * pa + 3 * k + ka] * p.pLam[(size_t)(15 * pa + k) * d.np + 15 * pb + m] *
                                 p.pJq[9 * pb + 3 * m + kb];
                } else if (ka < 3) {
                    for (int k = 0; k < 3; ++k) v += p.pJq[9 * pa + 3 * k + ka] * p.pLam[(size_t)(15 * pa + k) * d.np + 15 * pb + kb];
                } else if (kb < 3) {
                    for (int m = 0; m < 3; ++m) v += p.pLam[(size_t)(15 * pa + ka) * d.np + 15 * pb + m] * p.pJq[9 * pb + 3 * m + kb];
                } else {
                    v = p.pLam[(size_t)(15 * pa + ka) * d.np + 15 * pb + kb];
                }
                h += v;
            }
        }
        if (!bact) h = 0.0;
    }
    p.Hpp[e] = h;
    if (want_g) p.gp[a] = g;
}
__global__ __launch_bounds__(256) void kb_assemble(BaDims d, BaPtrs p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < d.n * d.n) assemble_item(d, p, e);
}

// ------------------------------------------------------------------ preparation
// Jacobi scales (first linearisation only), dogleg diagonal, scaled gradient, landmark Schur weights.
__device__ __forceinline__ void prepare_block(const BaDims &d, const BaPtrs &p) {
    const BaCtl *c = p.ctl;
    const double mu = c->mu;
    for (int a = threadIdx.x; a < d.n; a += blockDim.x) {
        const double h = p.Hpp[(size_t)a * d.n + a];
        if (c->first) p.sp[a] = 1.0 / (1.0 + sqrt(h));
        const double s = p.sp[a];
        p.diagD[a] = sqrt(fmin(fmax(s * s * h, 1e-6), 1e32));
        p.gs[a] = s * p.gp[a];
    }
    for (int l = threadIdx.x; l < d.L; l += blockDim.x) {
        const double h = p.lact[l] ? p.hll[l] : 0.0;   // constant landmarks take no part; keep their scales finite
        if (c->first) p.sl[l] = 1.0 / (1.0 + sqrt(h));
        const double s = p.sl[l];
        const double D = sqrt(fmin(fmax(s * s * h, 1e-6), 1e32));
        p.diagD[d.n + l] = D;
        p.gs[d.n + l] = p.lact[l] ? s * p.gl[l] : 0.0;
        p.omega[l] = p.lact[l] ? 1.0 / (h + mu * D * D / (s * s)) : 0.0;
    }
}
__global__ __launch_bounds__(256) void kb_prepare(BaDims d, BaPtrs p) { prepare_block(d, p); }

// ------------------------------------------------------------------- MFMA Schur
// T = W^T diag(omega) W, i.e. H_pl H_ll'^-1 H_lp of the reduced camera system, on the f64 matrix
// cores.  One workgroup (4 wavefronts) per 16x16 output tile; the landmark (K) dimension is split
// across the 4 wavefronts and combined in LDS in a fixed order.
// v_mfma_f64_16x16x4_f64 operands: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15];
// D: col = lane&15, row = (lane>>4) + 4*reg.
typedef double double4_t __attribute__((ext_vector_type(4)));

// emit_s: also write the pose-pose entries of the reduced system  S = sp (Hpp - T) sp + mu D^2  (lower triangle,
// packed over the active dofs) into Sred -- the solve kernel then only copies a contiguous block into LDS.
// where entry (i, j), j <= i, of the reduced system goes in Sred
__device__ __forceinline__ int sred_idx(const BaDims &d, int i, int j) { return d.sred_tiled ? tl_idx(i, j) : tri_idx(i, j); }
// blocks of 256 entries that cover what the "rest" blocks write: the na x na square, or -- tiled -- the padded square of the tiles
// (rows / columns >= na: identity padding and zeros, written every round: the factorisation runs in place)
__host__ __device__ __forceinline__ int sred_rest_blocks(const BaDims &d) {
    const int side = d.sred_tiled ? 16 * tl_tile_rows(d.na + 1) : d.na;
    return (side * side + 255) / 256;
}
__device__ __forceinline__ void schur_tile_block(const BaDims &d, const BaPtrs &p, int tile, bool emit_s = false) {
    __shared__ double red[4][256];
    const int tiles = d.PF / 16;
    const int ti = tile / tiles, tj = tile - ti * tiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = d.Lp / 4;   // landmarks per wavefront (multiple of 4)
    const int k0 = wave * kq;
    const int i = lane & 15, kk = lane >> 4;
    if (d.schur_mode == 0) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        // six operand pairs (18 loads) in flight per trip: with one, each of the ~18 trips waited out an L2 round trip before its
        // matrix-core instruction (the products are issued in the same order: the same bits)
        constexpr int SU = 6;
        for (int k = k0; k < k0 + kq; k += 4 * SU) {
            double av[SU], bv[SU], wv[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int l = min(k + 4 * u + kk, d.Lp - 1);
                wv[u] = (l < d.L) ? p.omega[l] : 0.0;
                av[u] = p.Wt[(size_t)l * d.PF + 16 * ti + i];
                bv[u] = p.Wt[(size_t)l * d.PF + 16 * tj + i];
            }
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (k + 4 * u < k0 + kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u] * wv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(kk + 4 * r) * 16 + i] = acc[r];
    } else {
        // BASELINE config 5's study ("fp32 vs bf16 BA solve"): the same contraction with f32 operands and accumulation
        // (v_mfma_f32_16x16x4_f32) or bf16 operands with f32 accumulation (v_mfma_f32_16x16x16_bf16) inside a real solve.  Accumulator
        // layout of both: register r of lane (i, kk) is element (row 4 kk + r, column i).
        typedef float f4_t __attribute__((ext_vector_type(4)));
        typedef short s4_t __attribute__((ext_vector_type(4)));
        f4_t acc = {0.f, 0.f, 0.f, 0.f};
        auto bf16 = [](float f) -> short {   // round to nearest even
            unsigned u = __float_as_uint(f);
            u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
            return (short)u;
        };
        if (d.schur_mode == 1) {
            for (int k = k0; k < k0 + kq; k += 4) {
                const int l = k + kk;
                const double w = (l < d.L) ? p.omega[l] : 0.0;
                const float a = (float)p.Wt[(size_t)l * d.PF + 16 * ti + i];
                const float b = (float)(p.Wt[(size_t)l * d.PF + 16 * tj + i] * w);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        } else {
            for (int k = k0; k < k0 + kq; k += 16) {   // K = 16: lane (i, kk) carries landmarks k + 4 kk .. + 3
                s4_t a, b;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int l = k + 4 * kk + q;
                    const bool in = l < k0 + kq;
                    const double w = (in && l < d.L) ? p.omega[l] : 0.0;
                    const size_t row = (size_t)(in ? l : k0) * d.PF;
                    a[q] = in ? bf16((float)p.Wt[row + 16 * ti + i]) : (short)0;
                    b[q] = in ? bf16((float)(p.Wt[row + 16 * tj + i] * w)) : (short)0;
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * kk + r) * 16 + i] = (double)acc[r];
    }
    __syncthreads();
    const int e = threadIdx.x;   // 256 elements of the tile
    const double s = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    const int row = e >> 4, col = e & 15;
    p.T[(size_t)(16 * ti + row) * d.PF + 16 * tj + col] = s;
    if (emit_s) {
        const int pa = 16 * ti + row, pb = 16 * tj + col;
        if (pa < 6 * d.F && pb < 6 * d.F) {
            const int a = 15 * (pa / 6) + pa % 6, b = 15 * (pb / 6) + pb % 6;
            const int ia = p.act_inv[a], ib = p.act_inv[b];
            if (ia >= 0 && ib >= 0 && ib <= ia) {
                double v = (p.Hpp[(size_t)a * d.n + b] - s) * (p.sp[a] * p.sp[b]);
                if (a == b) v += p.ctl->mu * p.diagD[a] * p.diagD[a];
                p.Sred[sred_idx(d, ia, ib)] = v;
            }
        }
    }
}
// the entries of S that carry no Schur term (a velocity / bias dof on either side, or no free landmark at all)
__device__ __forceinline__ void reduced_rest_block(const BaDims &d, const BaPtrs &p, int blk) {
    const int e = blk * 256 + (int)threadIdx.x;
    const int side = d.sred_tiled ? 16 * tl_tile_rows(d.na + 1) : d.na;
    if (e >= side * side) return;
    const int i = e / side, j = e - i * side;
    if (j > i) {   // above the diagonal: only the diagonal tiles have such slots (never read by another lane; kept finite)
        if (d.sred_tiled && (i >> 4) == (j >> 4)) p.Sred[tl_tile(i >> 4, i >> 4) + (j & 15) * TL_LD + (i & 15)] = 0.0;
        return;
    }
    if (i >= d.na) {   // tiled only: padding rows (the right-hand side is put into row na by the solve itself)
        p.Sred[tl_idx(i, j)] = (i == j) ? 1.0 : 0.0;
        return;
    }
    const int a = p.act_idx[i], b = p.act_idx[j];
    const int ka = a % 15, kb = b % 15;
    if (d.nla && ka < 6 && kb < 6) return;   // written by the Schur tile that owns it
    double v = p.Hpp[(size_t)a * d.n + b] * (p.sp[a] * p.sp[b]);
    if (a == b) v += p.ctl->mu * p.diagD[a] * p.diagD[a];
    p.Sred[sred_idx(d, i, j)] = v;
}
__global__ __launch_bounds__(256) void kb_schur_mfma(BaDims d, BaPtrs p) { schur_tile_block(d, p, blockIdx.x); }

// -------------------------------------------------------------------- the solve
// Dogleg needs three quadratic forms of the Jacobi-scaled Hessian Hs = J^T J (frames + landmarks) per linearisation:
// with g~ = grad / D = gs / D^2 (steepest descent) and n~ = gn / D (Gauss-Newton, n~ = -y, (Hs + mu D^2) y = gs)
//   Q(g~,g~)  the Cauchy step length;  and for a dogleg point  step = ca g~ + cb n~
//   step^T Hs step = ca^2 Q(g~,g~) + 2 ca cb Q(g~,n~) + cb^2 Q(n~,n~).
// Only Q(g~,g~) needs the matrix; the other two follow from the linear system the Gauss-Newton step solves:
//   Hs n~ = -gs - mu D^2 n~   =>   Q(g~,n~) = -|grad|^2 - mu (n~ . gs),   Q(n~,n~) = -(n~ . gs) - mu |gn|^2.
// kb_solve_aux is the wide (multi-workgroup) companion of the single-workgroup kb_solve; it runs before it and
// does the two passes over the big operands that do not depend on the factorisation:
//   role 0 (blocks [0, nbq)):       partial sums of Q(g~,g~): 16 frame rows or 32 landmark rows per block
//   role 1 (blocks [nbq, nbq+nbr)): wog = W^T (omega gl), 64 pose columns per block
__host__ __device__ __forceinline__ int aux_quad_blocks_n(int n, int L) { return (n + 15) / 16 + (L + 31) / 32; }
constexpr int WOG_CH = 8;   // landmark-row chunks of the W^T (omega gl) product (one workgroup each per 64 pose columns)
__host__ __device__ __forceinline__ int aux_wog_blocks(int F) { return WOG_CH * ((6 * F + 63) / 64); }

__device__ __forceinline__ double wog_at(const BaDims &d, const BaPtrs &p, int c) {   // the chunks' partial sums, in chunk order
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < WOG_CH; ++ch) s += p.wog[(size_t)ch * d.PF + c];
    return s;
}
__device__ __forceinline__ void solve_aux_block(const BaDims &d, const BaPtrs &p, int blk) {
    __shared__ double scratch[8];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = d.n, P6 = 6 * d.F;
    const int nbf = (n + 15) / 16, nbq = aux_quad_blocks_n(n, d.L);
    if (blk < nbq) {
        double acc = 0;
        if (blk < nbf) {   // frame rows a0 .. a0+3 of this wavefront against all frame columns
            const int a0 = 16 * blk + 4 * wave;
            double ga[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + r;
                ga[r] = 0.0;
                if (a < n && dof_active(p.fix, a)) {
                    const double D = p.diagD[a];
                    ga[r] = p.sp[a] * (p.gs[a] / (D * D));
                }
            }
            for (int b0 = 0; b0 < n; b0 += 256) {
                double h[4][4], xb[4];
#pragma unroll
                for (int cch = 0; cch < 4; ++cch) {
                    const int b = b0 + 64 * cch + lane;
                    xb[cch] = 0.0;
                    if (b < n && dof_active(p.fix, b)) {
                        const double D = p.diagD[b];
                        xb[cch] = p.sp[b] * (p.gs[b] / (D * D));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[cch][r] = (b < n && a0 + r < n) ? p.Hpp[(size_t)(a0 + r) * n + b] : 0.0;
                }
#pragma unroll
                for (int cch = 0; cch < 4; ++cch) {
                    double t = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += ga[r] * h[cch][r];
                    acc += t * xb[cch];
                }
            }
        } else {   // landmark rows: 2 g~_l sl (W_l . u6) + (g~_l sl)^2 hll
            const int l0 = 32 * (blk - nbf) + 8 * wave;
            double wg[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) wg[r] = 0.0;
            for (int c0 = 0; c0 < P6; c0 += 128) {
                double w[2][8], xc[2];
#pragma unroll
                for (int cch = 0; cch < 2; ++cch) {
                    const int cc = c0 + 64 * cch + lane;
                    xc[cch] = 0.0;
                    if (cc < P6) {
                        const int f = cc / 6, a = 15 * f + (cc - 6 * f);
                        if (pose_free(p.fix[f])) {
                            const double D = p.diagD[a];
                            xc[cch] = p.sp[a] * (p.gs[a] / (D * D));
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        w[cch][r] = (cc < P6 && l0 + r < d.L) ? p.Wt[(size_t)(l0 + r) * d.PF + cc] : 0.0;
                }
#pragma unroll
                for (int cch = 0; cch < 2; ++cch)
#pragma unroll
                    for (int r = 0; r < 8; ++r) wg[r] += w[cch][r] * xc[cch];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int l = l0 + r;
                if (l >= d.L || !p.lact[l]) continue;
                const double D = p.diagD[n + l];
                const double gl = p.sl[l] * (p.gs[n + l] / (D * D));
                acc += 2.0 * gl * wg[r];
                if (lane == 0) acc += gl * gl * p.hll[l];
            }
        }
        acc = block_sum(acc, scratch);
        if (tid == 0) p.partial[blk] = acc;
        return;
    }
    blk -= nbq;
    {   // wog[c] = sum_l Wt[l][c] omega_l gl_l.  WOG_CH row chunks x 64 pose columns per block, the four wavefronts split the chunk's
        // landmark rows; the chunks' partial sums are added by the reader in chunk order (wog_at).  As ONE block per 64 columns --
        // 278 rows of a freshly written 180 KB operand behind one workgroup -- this role took 15-21 us, three times the tiles and the
        // quadratic-form blocks of the same launch (in-kernel block timers, round 3).
        const int ncol = (P6 + 63) / 64, ch = blk / ncol, cb = blk - ch * ncol;
        const int cc = 64 * cb + lane;
        const int perc = (d.L + WOG_CH - 1) / WOG_CH, c0 = min(d.L, ch * perc), c1 = min(d.L, c0 + perc);
        const int per = (c1 - c0 + 3) / 4, l0 = min(c1, c0 + wave * per), l1 = min(c1, l0 + per);
        double sacc = 0;
        constexpr int WB = 12;
        if (cc < P6)
            for (int lb = l0; lb < l1; lb += WB) {
                double wv[WB], og[WB];
#pragma unroll
                for (int r = 0; r < WB; ++r) {
                    const int l = min(lb + r, max(l1 - 1, 0));
                    wv[r] = p.Wt[(size_t)l * d.PF + cc];
                    og[r] = (lb + r < l1) ? p.omega[l] * p.gl[l] : 0.0;
                }
#pragma unroll
                for (int r = 0; r < WB; ++r) sacc += wv[r] * og[r];
            }
        part[wave][lane] = sacc;
        __syncthreads();
        if (wave == 0 && cc < P6) p.wog[(size_t)ch * d.PF + cc] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    }
}
__device__ __forceinline__ void gradmax_block(const BaDims &d, const BaPtrs &p, double *scratch);
__device__ __forceinline__ void sum_cost_block(const BaDims &d, const BaPtrs &p, double *scratch);
// Schur tiles and the solve's auxiliary passes in one launch: [tiles^2 | aux blocks]
// (without free landmarks there is no Schur complement and no W^T (omega gl): only the quadratic-form blocks run)
// layout of the grid: [Schur tiles | rest of S | aux blocks]
__device__ __forceinline__ void d_schur_aux(const BaDims &d, const BaPtrs &p, int bx, int gx) {
#ifdef XRHIP_KPROF_PRINT
    const long long t0 = wall_clock64();
    struct Tail {
        long long t0;
        const BaDims &d;
        const BaPtrs &p;
        int bx;
        __device__ ~Tail() {
            const int nrest = sred_rest_blocks(d), tiles = d.PF / 16, b = bx;
            const int nbq = aux_quad_blocks_n(d.n, d.L);
            if (threadIdx.x == 0 && d.M > 1500 &&
                (b == 1 || b == nrest + 3 || b == nrest + tiles * tiles + 1 || b == nrest + tiles * tiles + nbq - 2 || b == nrest + tiles * tiles + nbq))
                printf("kb_schur_aux block %d (rest %d tiles %d quad %d): %lld x10ns\n", b, nrest, tiles * tiles, nbq, wall_clock64() - t0);
        }
    } tail{t0, d, p, bx};
#endif
    const int t2 = d.nla ? (d.PF / 16) * (d.PF / 16) : 0;
    const int nrest = sred_rest_blocks(d);
    int blk = bx;
    if (blk == gx - 1) {
        // Round 5: the linearisation's total cost and gradient max-norm -- two block-wide reductions and an exponential map per frame
        // that nothing in this launch reads (kb_solve_try does) -- ride here as one more block beside the tiles instead of in front of
        // them in the single-workgroup kb_cost_prepare (11 us per round, of which the preparation the tiles DO need is the smaller half:
        // kb_prepare).  Same functions, same bits.
        __shared__ double scratch[8];
        sum_cost_block(d, p, scratch);
        __syncthreads();
        gradmax_block(d, p, scratch);
        return;
    }
    if (blk < t2) {
        schur_tile_block(d, p, blk, true);
        return;
    }
    blk -= t2;
    if (blk < nrest) {
        reduced_rest_block(d, p, blk);
        return;
    }
    solve_aux_block(d, p, blk - nrest);
}
__global__ __launch_bounds__(256) void kb_schur_aux(BaDims d, BaPtrs p) { d_schur_aux(d, p, blockIdx.x, gridDim.x); }

// Reduced camera system + blocked Cholesky + Gauss-Newton / Cauchy quantities.  One workgroup.
// Only the `na` free frame dofs enter the factorisation (localize_newframe has 15, refine_window 15 F).
// Dynamic LDS: rhs [na] + packed lower triangle when it fits (use_lds); otherwise the triangle lives in Sred.
__device__ __forceinline__ void solve_block(const BaDims &d, const BaPtrs &p, int use_lds, double *lds) {
    __shared__ double scratch[32];
    __shared__ double Dblk[CH_NB][CH_NB + 1];
    __shared__ int fail;
    BaCtl *c = p.ctl;
    const int n = d.n, na = d.na, tid = threadIdx.x, nt = blockDim.x;
    double *work = lds;                                // LDS work region: the packed triangle, later the gathered frame step
    const double mu = c->mu;
    KPROF_BEGIN();
    int bad = 0;
    // The factorisation works on `A`: the packed lower triangle (compact) + the rhs as its row `na`, in LDS when it
    // fits, else in Sred.  As ONE pointer (`use_lds ? work : p.Sred`) it is a generic one and every access of the
    // Cholesky / substitution chain becomes a FLAT instruction even when it goes to LDS (ISA of v17: 244 flat_load /
    // flat_store in kb_solve_try); here the stage is instantiated once per memory, so the LDS instance reads and
    // writes with ds_* instructions.  Returns false when the factorisation failed.
    auto factor_stage = [&](double *A) __attribute__((always_inline)) -> bool {
        double *y = A + tri_idx(na, 0);                    // [na] rhs -> L^-1 rhs (by the factorisation) -> solution
        for (int i = tid; i < na; i += nt) {   // rhs = sp (gp - W^T (omega gl)); the product comes from kb_solve_aux
            const int a = p.act_idx[i];
            const int fa = a / 15, ka = a - 15 * fa;
            const double sacc = (d.nla && ka < 6) ? wog_at(d, p, 6 * fa + ka) : 0.0;
            y[i] = (p.gp[a] - sacc) * p.sp[a];
        }
        __syncthreads();
        KPROF(0);
        // ---- S = sp (Hpp - T) sp + mu D^2 over the free dofs was written by kb_schur_aux as a packed triangle in Sred
        if (use_lds) {   // eight loads per thread in flight (one at a time, this copy took 7 us at 165 unknowns)
            const int tri = na * (na + 1) / 2;
            const double *src = p.Sred;
            for (int e0 = tid; e0 < tri; e0 += 8 * nt) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = src[min(e0 + q * nt, tri - 1)];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (e0 + q * nt < tri) A[e0 + q * nt] = v[q];
            }
        }
        __syncthreads();
        KPROF(1);
#ifdef XRHIP_KPROF
        const bool ok = chol_blocked(A, na, na + 1, Dblk, &fail, p.ctl->prof + 24);   // the rhs row rides along
#else
        const bool ok = chol_blocked(A, na, na + 1, Dblk, &fail);
#endif
        KPROF(2);
        if (!ok) {
            if (tid == 0) c->linear_ok = 0;
            return false;
        }
        KPROF(3);
        trsv_lower_t(A, na, y);
        KPROF(4);
        // ---- Gauss-Newton step (scaled space), landmark back-substitution, dogleg gradient
        for (int a = tid; a < n; a += nt) {
            p.gn[a] = 0.0;
            p.grad[a] = dof_active(p.fix, a) ? p.gs[a] / p.diagD[a] : 0.0;
            p.delta[a] = 0.0;   // delta doubles as the full-layout y (frame part) for the back-substitution below
        }
        __syncthreads();
        for (int i = tid; i < na; i += nt) {
            const int a = p.act_idx[i];
            const double ya = y[i];
            p.gn[a] = -p.diagD[a] * ya;
            p.delta[a] = ya;
            if (!isfinite(ya)) bad = 1;
        }
        __syncthreads();
        return true;
    };
    // use_lds == 2: the tiled layout of dense_lds.hip.h (bank-conflict-free 16x16 tiles, block inverses kept for the
    // back-substitution).  The reduced system arrives as a packed triangle in Sred; it is dealt to the tiles by wavefront -- lane
    // (column c = lane & 15, row rr + 4 pass): sixteen lanes read 128 contiguous bytes of one row.
    // in_place (round 6): the system does not fit LDS -- kb_schur_aux wrote it into Sred in the tiled layout, padding included
    // (BaDims::sred_tiled), and it is factored there, the tiles served by L2 (rounds 1-5: the packed routines on the global buffer,
    // 314 us at 240 unknowns); only L^-1 rhs / the solution live in LDS.
    auto factor_stage_tiled = [&](double *A, bool in_place) __attribute__((always_inline)) -> bool {
        const int nrows = na + 1, T = tl_tile_rows(nrows);
        double *yv = in_place ? work : A + tl_doubles(nrows);
        const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6, cc = lane & 15, rr = lane >> 4;
        const int n_tiles = T * (T + 1) / 2;
        const double *src = p.Sred;
        KPROF(0);
        // the right-hand side row sp (gp - W^T (omega gl)): its two-level gathers are issued first and parked in a register
        double rhs_v = 0.0;
        if (tid < na) {
            const int a = p.act_idx[tid];
            const int fa = a / 15, ka = a - 15 * fa;
            const double sacc = (d.nla && ka < 6) ? wog_at(d, p, 6 * fa + ka) : 0.0;
            rhs_v = (p.gp[a] - sacc) * p.sp[a];
        }
        constexpr int TU = 3;   // tiles (4 loads each) in flight per wavefront; loads are unconditional (clamped), selected afterwards
        for (int t0 = wave; t0 < (in_place ? 0 : n_tiles); t0 += TU * nw) {
            double v[TU][4];
            int base[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int t = min(t0 + u * nw, n_tiles - 1);
                int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
                while (ti * (ti + 1) / 2 > t) --ti;
                while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                const int tj = t - ti * (ti + 1) / 2;
                base[u] = tl_tile(ti, tj) + cc * TL_LD + rr;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int i = 16 * ti + rr + 4 * ps, j = 16 * tj + cc;
                    const bool in = i < na && j <= i;
                    const double x = src[in ? tri_idx(i, j) : 0];
                    v[u][ps] = in ? x : ((i == j && i >= na) ? 1.0 : 0.0);   // identity padding; zero elsewhere outside the triangle
                }
            }
#pragma unroll
            for (int u = 0; u < TU; ++u)
                if (t0 + u * nw < n_tiles)
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) A[base[u] + 4 * ps] = v[u][ps];
        }
        __syncthreads();
        if (tid < na) A[tl_idx(na, tid)] = rhs_v;
        __syncthreads();
        KPROF(1);
#ifdef XRHIP_KPROF
        const bool ok = tl_chol(A, na, nrows, &Dblk[0][0], &fail, p.ctl->prof + 24);
#else
        const bool ok = tl_chol(A, na, nrows, &Dblk[0][0], &fail);
#endif
        KPROF(2);
        if (!ok) {
            if (tid == 0) c->linear_ok = 0;
            return false;
        }
        for (int i = tid; i < 16 * T; i += nt) yv[i] = i < na ? A[tl_idx(na, i)] : 0.0;   // L^-1 rhs; zero beyond n (tl_trsv_t)
        __syncthreads();
        KPROF(3);
        tl_trsv_t(A, na, yv);
        KPROF(4);
        for (int a = tid; a < n; a += nt) {
            p.gn[a] = 0.0;
            p.grad[a] = dof_active(p.fix, a) ? p.gs[a] / p.diagD[a] : 0.0;
            p.delta[a] = 0.0;
        }
        __syncthreads();
        for (int i = tid; i < na; i += nt) {
            const int a = p.act_idx[i];
            const double ya = yv[i];
            p.gn[a] = -p.diagD[a] * ya;
            p.delta[a] = ya;
            if (!isfinite(ya)) bad = 1;
        }
        __syncthreads();
        return true;
    };
    if (use_lds == 2) {
        if (!factor_stage_tiled(work, false)) return;
    } else if (use_lds) {
        if (!factor_stage(work)) return;
    } else if (d.sred_tiled) {
        if (!factor_stage_tiled(static_cast<double *>(p.Sred), true)) return;
    } else {
        if (!factor_stage(static_cast<double *>(p.Sred))) return;
    }
    if (!d.nla) {   // every landmark is held constant (localize_newframe, refine_subwindow): nothing to back-substitute
        for (int l = tid; l < d.L; l += nt) {
            p.gn[n + l] = 0.0;
            p.grad[n + l] = 0.0;
        }
    } else {
        const int P6 = 6 * d.F;
        double *x6 = work;   // [P6] scaled frame step gathered to the pose columns (the triangle is dead by now)
        for (int cidx = tid; cidx < P6; cidx += nt) {
            const int f = cidx / 6, k = cidx - 6 * f;
            x6[cidx] = p.sp[15 * f + k] * p.delta[15 * f + k];
        }
        __syncthreads();
        // One landmark per thread: its row of W against the frame step, a plain 6F-term dot product (x6 is a broadcast
        // read from LDS, the row streams through the thread's own cache lines).  The first version gave a wavefront
        // eight rows and reduced every row across the lanes -- eight butterfly reductions per pass, five passes for 278
        // landmarks, 35 us of the 158 us this kernel took on a 10-keyframe window (profiles/r02_kprof_v19.md).
        for (int l = tid; l < d.L; l += nt) {
            double w = 0.0;
            if (p.lact[l]) {
                const double *row = p.Wt + (size_t)l * d.PF;
#pragma unroll 8
                for (int cidx = 0; cidx < P6; ++cidx) w += row[cidx] * x6[cidx];
            }
            double yl = 0.0;
            const double D = p.diagD[n + l];
            if (p.lact[l]) {
                const double sl = p.sl[l];
                yl = (sl * p.gl[l] - sl * w) / (sl * sl * p.hll[l] + mu * D * D);
                if (!isfinite(yl)) bad = 1;
            }
            p.gn[n + l] = -D * yl;
            p.grad[n + l] = p.lact[l] ? p.gs[n + l] / D : 0.0;
        }
    }
    if (bad) atomicExch(&fail, 1);
    __syncthreads();
    if (fail) {
        if (tid == 0) c->linear_ok = 0;
        return;
    }
    KPROF(5);
    // ---- Cauchy point alpha = |grad|^2 / Q(g~,g~) and the quadratic forms of the dogleg model (see kb_solve_aux)
    double r3[3] = {0, 0, 0};   // |grad|^2, n~ . gs, |gn|^2
    for (int a = tid; a < d.NV; a += nt) {
        const double g = p.grad[a], nn = p.gn[a];
        r3[0] += g * g;
        r3[1] += (nn / p.diagD[a]) * p.gs[a];
        r3[2] += nn * nn;
    }
    block_sum_n<3>(r3, scratch);
    KPROF(6);
    if (tid == 0) {
        double qgg = 0;
        const int nbq = aux_quad_blocks_n(d.n, d.L);
        for (int i = 0; i < nbq; ++i) qgg += p.partial[i];
        c->alpha = r3[0] / qgg;
        c->q_gg = qgg;
        c->q_gn = -r3[0] - mu * r3[1];
        c->q_nn = -r3[1] - mu * r3[2];
        c->linear_ok = 1;
    }
}

// gradient_max_norm = |x - Plus(x, -g)|_inf in ambient coordinates (TrustRegionMinimizer::EvaluateGradientAndJacobian)
__device__ __forceinline__ void gradmax_block(const BaDims &d, const BaPtrs &p, double *scratch) {   // scratch: blockDim/64 doubles
    double m = 0;
    for (int f = threadIdx.x; f < d.F; f += blockDim.x) {
        double neg[15], out[16];
        for (int k = 0; k < 15; ++k) neg[k] = -p.gp[15 * f + k];
        const double *s = p.state + 16 * f;
        state_plus(s, neg, pose_free(p.fix[f]), motion_free(p.fix[f]), out);
        for (int k = 0; k < 16; ++k) m = fmax(m, fabs(s[k] - out[k]));
    }
    for (int l = threadIdx.x; l < d.L; l += blockDim.x)
        if (p.lact[l]) m = fmax(m, fabs(p.gl[l]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
        for (int i = 0; i < nw; ++i) r = fmax(r, scratch[i]);
        p.ctl->gmax = r;
    }
    __syncthreads();
}

// total cost at `state` (x) from the per-factor costs written by the linearisation kernels
__device__ __forceinline__ void sum_cost_block(const BaDims &d, const BaPtrs &p, double *scratch) {
    double c = 0;
    for (int o = threadIdx.x; o < d.M; o += blockDim.x) c += p.ocost[o];
    for (int o = threadIdx.x; o < d.MR; o += blockDim.x) c += p.rcost[o];
    for (int k = threadIdx.x; k < d.NI; k += blockDim.x) c += p.imu_cost[k];
    c = block_sum(c, scratch);
    if (threadIdx.x == 0) {
        c += p.pcost[0];
        if (d.np)
            for (int b = 0; b < prior_row_blocks(d.np); ++b) c += p.pcost[1 + b];   // row-block form of the prior (zeros otherwise)
        BaCtl *ctl = p.ctl;
        ctl->x_cost = c;
        if (ctl->first) {
            ctl->initial_cost = c;
            ctl->minimum_cost = c;
        }
    }
}
__global__ __launch_bounds__(256) void kb_sum_cost(BaDims d, BaPtrs p) {
    __shared__ double scratch[8];
    sum_cost_block(d, p, scratch);
}

// ambient norm of the active parameter blocks of `state`/`depth`; block-wide
__device__ __forceinline__ double ambient_norm2(const BaDims &d, const BaPtrs &p, const double *state,
                                                const double *depth, double *scratch) {
    double s = 0;
    for (int f = threadIdx.x; f < d.F; f += blockDim.x) {
        const double *x = state + 16 * f;
        if (pose_free(p.fix[f]))
            for (int k = 0; k < 7; ++k) s += x[k] * x[k];
        if (motion_free(p.fix[f]))
            for (int k = 7; k < 16; ++k) s += x[k] * x[k];
    }
    for (int l = threadIdx.x; l < d.L; l += blockDim.x)
        if (p.lact[l]) s += depth[l] * depth[l];
    return block_sum(s, scratch);
}

// ------------------------------------------------------------------- trial logic
// TrustRegionMinimizer's scalars.  Every thread keeps its own copy and runs the (uniform) decision logic itself;
// thread 0 writes them back.  Shared by the in-kernel trial loop (try_block) and by kb_trials_wide.
struct TrialScalars {
    int iteration, invalid_steps, successful_steps, reuse, termination, status, max_iterations, linear_ok;
    double radius, mu, cand_cost, last_step_norm;
    double gmax, x_cost, x_norm, alpha, q_gg, q_gn, q_nn, gnorm, gn_norm, gd;
};
constexpr double TR_FUNCTION_TOLERANCE = 1e-6, TR_GRADIENT_TOLERANCE = 1e-10, TR_PARAMETER_TOLERANCE = 1e-8;
constexpr double TR_MIN_RELATIVE_DECREASE = 1e-3, TR_MIN_RADIUS = 1e-32, TR_MAX_RADIUS = 1e16;

__device__ __forceinline__ void trial_load(const BaCtl *c, TrialScalars &t) {
    t.iteration = c->iteration;
    t.invalid_steps = c->invalid_steps;
    t.successful_steps = c->successful_steps;
    t.reuse = c->reuse;
    t.termination = c->termination;
    t.status = ST_RUNNING;
    t.max_iterations = c->max_iterations;
    t.linear_ok = c->linear_ok;
    t.radius = c->radius;
    t.mu = c->mu;
    t.cand_cost = c->cand_cost;
    t.last_step_norm = c->step_norm;
    t.gmax = c->gmax;
    t.x_cost = c->x_cost;
    t.x_norm = c->x_norm;
    t.alpha = c->alpha;
    t.q_gg = c->q_gg;
    t.q_gn = c->q_gn;
    t.q_nn = c->q_nn;
    t.gnorm = c->gnorm;
    t.gn_norm = c->gn_norm;
    t.gd = c->gd;
}
__device__ __forceinline__ void trial_store(BaCtl *c, const TrialScalars &t) {   // one thread
    c->iteration = t.iteration;
    c->invalid_steps = t.invalid_steps;
    c->successful_steps = t.successful_steps;
    c->reuse = t.reuse;
    c->termination = t.termination;
    c->radius = t.radius;
    c->mu = t.mu;
    c->cand_cost = t.cand_cost;
    c->step_norm = t.last_step_norm;
}

// FinalizeIterationAndCheckIfMinimizerCanContinue + start of the next iteration, for the first trial of a launch
__device__ __forceinline__ void trial_begin(TrialScalars &t, bool skip_finalize, bool check_gradient) {
    if (!skip_finalize) {
        if (t.iteration >= t.max_iterations) {
            t.termination = XRHIP_BA_NO_CONVERGENCE;
            t.status = ST_DONE;
        } else if (check_gradient && t.gmax <= TR_GRADIENT_TOLERANCE) {
            t.termination = XRHIP_BA_CONVERGENCE;
            t.status = ST_DONE;
        } else if (t.radius <= TR_MIN_RADIUS) {
            t.termination = XRHIP_BA_CONVERGENCE;
            t.status = ST_DONE;
        }
        if (t.status == ST_RUNNING) t.iteration += 1;
    }
    if (t.status == ST_RUNNING && !t.linear_ok) {
        if (t.mu * 10.0 < 1.0) {
            t.mu *= 10.0;           // ComputeGaussNewtonStep: retry with a larger mu, same iteration
            t.status = ST_RESOLVE_INNER;
        } else {
            t.invalid_steps += 1;   // LINEAR_SOLVER_FAILURE -> invalid step
            if (t.invalid_steps >= 5) {
                t.termination = XRHIP_BA_FAILURE;
                t.status = ST_DONE;
            } else {
                t.mu *= 10.0;
                t.reuse = 0;
                t.status = ST_RESOLVE;
            }
        }
    }
}

// traditional dogleg point for trust-region radius rk: step (scaled by D) = ca grad + cb gn.
// step_norm < 0: the interpolated case, the caller supplies |step| from its reduction.
__device__ __forceinline__ void dogleg_point(const TrialScalars &t, double rk, double &ca, double &cb, double &step_norm) {
    ca = 0.0;
    cb = 0.0;
    if (t.gn_norm <= rk) {
        cb = 1.0;
        step_norm = t.gn_norm;
    } else if (t.gnorm * t.alpha >= rk) {
        ca = -(rk / t.gnorm);
        step_norm = rk;
    } else {
        // (DoglegStrategy::ComputeTraditionalDoglegStep writes these squares as pow(x, 2.0))
        const double b_dot_a = -t.alpha * t.gd;
        const double a_sq = (t.alpha * t.gnorm) * (t.alpha * t.gnorm);
        const double bma_sq = a_sq - 2 * b_dot_a + t.gn_norm * t.gn_norm;
        const double cc = b_dot_a - a_sq;
        const double dd = sqrt(cc * cc + bma_sq * (rk * rk - a_sq));
        const double beta = (cc <= 0) ? (dd - cc) / bma_sq : (rk * rk - a_sq) / (dd + cc);
        ca = -t.alpha * (1.0 - beta);
        cb = beta;
        step_norm = -1.0;
    }
}
__device__ __forceinline__ double dogleg_model_change(const TrialScalars &t, double ca, double cb, double step_dot_gs) {
    const double shs = (ca * ca) * t.q_gg + 2.0 * (ca * cb) * t.q_gn + (cb * cb) * t.q_nn;   // see kb_schur_aux
    return -step_dot_gs - 0.5 * shs;
}

// Decision for trial k of a batch whose predecessors were all rejected (k > 0 first replays their finalize step).
// Returns true when the candidate is accepted; t.status != ST_RUNNING ends the batch.
// The arithmetic of a decision that does not depend on the radius / counters the rejections before it changed -- the
// parameter-tolerance norm and the relative decrease: a square root and a division -- is split off (trial_terms), so that a
// batch of candidates computes it side by side, once per candidate, before walking through the decisions (kb_trials_wide:
// 15-25 decisions per launch on the reference's rejection runs, each of which used to repeat them).
struct TrialTerms {
    double cost, dn, relative_decrease;
};
__device__ __forceinline__ TrialTerms trial_terms(const TrialScalars &t, double model_cost_change, double cost, double dn2) {
    TrialTerms w;
    if (!isfinite(cost)) cost = 1.7976931348623157e308;
    w.cost = cost;
    w.dn = sqrt(dn2);
    w.relative_decrease = (t.x_cost - cost) / model_cost_change;
    return w;
}
__device__ __forceinline__ bool trial_decide(TrialScalars &t, int k, double model_cost_change, const TrialTerms &w, double step_norm) {
    if (k > 0) {
        if (t.iteration >= t.max_iterations) {
            t.termination = XRHIP_BA_NO_CONVERGENCE;
            t.status = ST_DONE;
        } else if (t.radius <= TR_MIN_RADIUS) {
            t.termination = XRHIP_BA_CONVERGENCE;
            t.status = ST_DONE;
        }
        if (t.status != ST_RUNNING) return false;
        t.iteration += 1;
    }
    if (!(model_cost_change > 0.0)) {
        t.invalid_steps += 1;
        if (t.invalid_steps >= 5) {
            t.termination = XRHIP_BA_FAILURE;
            t.status = ST_DONE;
        } else {
            t.mu *= 10.0;   // StepIsInvalid
            t.reuse = 0;
            t.status = ST_RESOLVE;
        }
        return false;
    }
    if (w.dn <= TR_PARAMETER_TOLERANCE * (t.x_norm + TR_PARAMETER_TOLERANCE) ||
        fabs(t.x_cost - w.cost) <= TR_FUNCTION_TOLERANCE * t.x_cost) {
        t.termination = XRHIP_BA_CONVERGENCE;
        t.status = ST_DONE;
        return false;
    }
    if (w.relative_decrease > TR_MIN_RELATIVE_DECREASE) {
        t.invalid_steps = 0;
        t.successful_steps += 1;
        if (w.relative_decrease < 0.25) t.radius *= 0.5;
        if (w.relative_decrease > 0.75) t.radius = fmax(t.radius, 3.0 * step_norm);
        t.radius = fmin(t.radius, TR_MAX_RADIUS);
        t.mu = fmax(1e-8, 2.0 * t.mu / 10.0);
        t.reuse = 0;
        t.cand_cost = w.cost;
        t.last_step_norm = step_norm;
        t.status = ST_ACCEPTED;
        return true;
    }
    t.invalid_steps = 0;
    t.radius *= 0.5;   // StepRejected
    t.reuse = 1;
    return false;
}
__device__ __forceinline__ bool trial_decide(TrialScalars &t, int k, double model_cost_change, double cost, double dn2,
                                             double step_norm) {
    return trial_decide(t, k, model_cost_change, trial_terms(t, model_cost_change, cost, dn2), step_norm);
}

// Publish the result of a trial launch into the host mailbox: control block, on termination the optimised
// states, last the sequence number.  Block-wide.
__device__ __forceinline__ void publish_block(const BaDims &d, const BaPtrs &p, int status, int seq, bool always) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (status == ST_DONE) {
        for (int e = tid; e < 16 * d.F; e += nt) p.host_out[e] = p.state[e];
        for (int l = tid; l < d.L; l += nt) p.host_out[16 * d.F + l] = p.depth[l];
        __threadfence_system();
    }
    if (tid == 0) p.ctl->status = status;
    __syncthreads();
    if (always || status == ST_DONE) {
        // one word per thread, read past this CU's vector cache (the block's own thread 0 has just written some of them; lines of the
        // control block are in the cache since trial_load).  As a loop of thread 0 -- 55 dependent load / host-store pairs -- the copy
        // took 7 us of every kb_trials_wide / kb_solve_try launch.
        constexpr unsigned ctl_words = sizeof(BaCtl) / sizeof(long long);
        long long *src = reinterpret_cast<long long *>(static_cast<BaCtl *>(p.ctl));
        long long *dst = reinterpret_cast<long long *>(static_cast<BaCtl *>(p.host_ctl));
        for (unsigned i = tid; i < ctl_words; i += nt) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __syncthreads();
        if (tid == 0) *reinterpret_cast<volatile int *>(static_cast<int *>(p.host_seq)) = seq;
    }
    __syncthreads();
}

// ------------------------------------------------------------------- trial loop
// TrustRegionMinimizer's per-iteration logic for as long as no new linearisation is needed:
// dogleg point for the current radius, model cost change, candidate = Plus(x, delta), candidate
// cost, tolerance checks, accept / reject / radius update.  One workgroup; exits with
//   ST_ACCEPTED  a step was accepted -> host relinearises at the new x
//   ST_RESOLVE   the step was invalid -> mu was increased, host re-solves the linear system
//   ST_DONE      the minimiser terminated
// `after_linearisation` = this launch directly follows a (re)linearisation or re-solve.
__device__ __forceinline__ int try_block(const BaDims &d, const BaPtrs &p, const Ext &cam, const Ext &imu, double sx,
                                         double sy, int after_linearisation, int seq, bool wide_after_first, double *sh,
                                         bool prep_only = false, bool publish_always = true) {
    // sh: LDS, TRY_B * np doubles for the prior deltas of the candidates + TRY_B * NI * 15 for the raw IMU residuals
    // wide_after_first: if the first trial is rejected, hand the following ones to kb_trials_wide (ST_NEED_TRIALS)
    __shared__ double scratch[2 * TRY_B * 8];
    BaCtl *c = p.ctl;
    const int tid = threadIdx.x, nt = blockDim.x;

    // mode 1: directly after a (re)linearisation at a new x (initial point or accepted step)
    // mode 2: after a re-solve caused by an invalid step (x unchanged)
    // mode 3: inner retry of DoglegStrategy::ComputeGaussNewtonStep's mu loop (same iteration)
    const int mode = after_linearisation;
    KPROF_BEGIN();
    if (mode == 1) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue of the iteration that produced this x:
        // refresh the user state the IMU factors read their bias reference from (StateUpdatingCallback)
        for (int k = tid; k < d.NI; k += nt) {
            const double *st = p.state + 16 * p.imu_i[k];
            for (int i = 0; i < 6; ++i) p.bias_ref[6 * k + i] = st[10 + i];
        }
        const double xn2 = ambient_norm2(d, p, p.state, p.depth, scratch);
        if (tid == 0) {
            c->x_norm = sqrt(xn2);
            c->first = 0;
        }
        __syncthreads();
    }
    KPROF(10);
    // |gradient|, |gauss-newton step| and their inner product stay the same for every trial of this linearisation
    {
        double r3[3] = {0, 0, 0};
        for (int a = tid; a < d.NV; a += nt) {
            r3[0] += p.grad[a] * p.grad[a];
            r3[1] += p.gn[a] * p.gn[a];
            r3[2] += p.grad[a] * p.gn[a];
        }
        block_sum_n<3>(r3, scratch);
        if (tid == 0) {
            c->gnorm = sqrt(r3[0]);
            c->gn_norm = sqrt(r3[1]);
            c->gd = r3[2];
        }
        __syncthreads();
    }
    KPROF(11);
    // prep_only: large problems cost every trial, the first included, on the whole chip (kb_trials_wide is already
    // queued behind this kernel); nothing is published from here
    if (prep_only) return ST_NEED_TRIALS;
    TrialScalars t;
    trial_load(c, t);
    bool check_gradient = (mode == 1);   // the iteration that led here was successful
    bool skip_finalize = (mode == 3);
    // Trials are evaluated in batches: after a rejection the next radii are known (radius / 2, / 4, ...), so the
    // next TRY_B candidates are costed in ONE sweep over the factors and the decisions replayed in order.  The
    // first trial of a launch is usually accepted and goes alone.
    int B = 1;
    while (t.status == ST_RUNNING) {
        trial_begin(t, skip_finalize, check_gradient);
        skip_finalize = false;
        check_gradient = false;
        if (t.status != ST_RUNNING) break;
        KPROF(12);
        // ---- traditional dogleg points for radius, radius/2, ...: step (scaled by D) = ca grad + cb gn
        double ca[TRY_B], cb[TRY_B], step_norm[TRY_B];
#pragma unroll
        for (int k = 0; k < TRY_B; ++k) {
            ca[k] = 0.0;
            cb[k] = 0.0;
            step_norm[k] = 0.0;
            if (k < B) dogleg_point(t, t.radius * (1.0 / (double)(1 << k)), ca[k], cb[k], step_norm[k]);
        }
        double red2[2 * TRY_B];   // per candidate: |step|^2 (D-scaled), step . gs
#pragma unroll
        for (int k = 0; k < 2 * TRY_B; ++k) red2[k] = 0.0;
        for (int a = tid; a < d.NV; a += nt) {
            const double g = p.grad[a], gnv = p.gn[a], D = p.diagD[a], gsa = p.gs[a];
            const double sc = a < d.n ? p.sp[a] : p.sl[a - d.n];
#pragma unroll
            for (int k = 0; k < TRY_B; ++k)
                if (k < B) {
                    const double v = ca[k] * g + cb[k] * gnv;
                    red2[2 * k] += v * v;
                    const double st = v / D;
                    red2[2 * k + 1] += st * gsa;
                    p.delta[(size_t)k * d.NV + a] = st * sc;
                }
        }
        if (B == 1) {
            double r1[2] = {red2[0], red2[1]};
            block_sum_n<2>(r1, scratch);
            red2[0] = r1[0];
            red2[1] = r1[1];
        } else {
            block_sum_n<2 * TRY_B>(red2, scratch);
        }
        double model_cost_change[TRY_B];
#pragma unroll
        for (int k = 0; k < TRY_B; ++k) {
            if (step_norm[k] < 0) step_norm[k] = sqrt(red2[2 * k]);
            model_cost_change[k] = dogleg_model_change(t, ca[k], cb[k], red2[2 * k + 1]);
        }
        __syncthreads();
        KPROF(13);
        // ---- candidate points and their costs
        for (int e = tid; e < B * d.F; e += nt) {
            const int k = e / d.F, f = e - k * d.F;
            state_plus(p.state + 16 * f, p.delta + (size_t)k * d.NV + 15 * f, pose_free(p.fix[f]), motion_free(p.fix[f]),
                       p.cand + (size_t)k * 16 * d.F + 16 * f);
        }
        for (int e = tid; e < B * d.L; e += nt) {
            const int k = e / d.L, l = e - k * d.L;
            p.depth_cand[(size_t)k * d.L + l] = p.depth[l] + (p.lact[l] ? p.delta[(size_t)k * d.NV + d.n + l] : 0.0);
        }
        __syncthreads();
        for (int e = tid; e < B * d.NP; e += nt) {   // prior delta at the candidates, staged for the row products
            const int k = e / d.NP, i = e - k * d.NP;
            double dl[15];
            prior_delta(p, i, p.cand + (size_t)k * 16 * d.F, dl, nullptr);
            for (int q = 0; q < 15; ++q) sh[k * d.np + 15 * i + q] = dl[q];
        }
        __syncthreads();
        KPROF(15);
        double red[2 * TRY_B];   // per candidate: cost, |x - candidate|^2
#pragma unroll
        for (int k = 0; k < 2 * TRY_B; ++k) red[k] = 0.0;
        for (int o = tid; o < d.M; o += nt)
#pragma unroll
            for (int k = 0; k < TRY_B; ++k)
                if (k < B)
                    red[2 * k] += obs_eval(d, p, o, p.cand + (size_t)k * 16 * d.F, p.depth_cand + (size_t)k * d.L, cam, sx, sy,
                                           false, nullptr);
        for (int o = tid; o < d.MR; o += nt)
#pragma unroll
            for (int k = 0; k < TRY_B; ++k)
                if (k < B) red[2 * k] += rot_eval(d, p, o, p.cand + (size_t)k * 16 * d.F, cam, sx, sy, false, nullptr);
        KPROF(16);
        {   // IMU factors: the SO(3)-heavy raw residual is a long serial chain, so every (factor, candidate) pair gets
            // its own lane (all pairs advance in lockstep); the 15x15 whitening then runs one row per thread
            double *raw = sh + TRY_B * d.np;   // [B * NI][15]
            for (int e = tid; e < B * d.NI; e += nt) {
                const int k = e / d.NI, f = e - k * d.NI;
                const int fi = p.imu_i[f], fj = p.imu_j[f];
                double r15[15];
                if (p.fix[fi] == 3 && p.fix[fj] == 3) {
                    for (int q = 0; q < 15; ++q) r15[q] = 0.0;
                } else {
                    const double *st = p.cand + (size_t)k * 16 * d.F;
                    imu_raw_residual(load_state(st + 16 * fi), load_state(st + 16 * fj), load_imu(p.imu_data + (size_t)f * XRHIP_IMU_DIM),
                                     v3(p.bias_ref[6 * f], p.bias_ref[6 * f + 1], p.bias_ref[6 * f + 2]),
                                     v3(p.bias_ref[6 * f + 3], p.bias_ref[6 * f + 4], p.bias_ref[6 * f + 5]), imu, r15);
                }
                for (int q = 0; q < 15; ++q) raw[15 * e + q] = r15[q];
            }
            __syncthreads();
            for (int it = tid; it < B * d.NI * 15; it += nt) {
                const int e = it / 15, i = it - 15 * e, k = e / d.NI, f = e - k * d.NI;
                const double *S = p.imu_data + (size_t)f * XRHIP_IMU_DIM + 56 + 15 * i;
                double acc = 0;
#pragma unroll
                for (int j = 0; j < 15; ++j) acc += S[j] * raw[15 * e + j];
                const double cst = 0.5 * acc * acc;
#pragma unroll
                for (int q = 0; q < TRY_B; ++q)
                    if (q == k) red[2 * q] += cst;
            }
            const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
            for (int i0 = 4 * wave; i0 < d.np; i0 += 4 * nw) {
                double s4[TRY_B][4];
#pragma unroll
                for (int k = 0; k < TRY_B; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s4[k][r] = 0.0;
                for (int j = lane; j < d.np; j += 64) {
                    double row[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[r] = p.pS[(size_t)min(i0 + r, d.np - 1) * d.np + j];
#pragma unroll
                    for (int k = 0; k < TRY_B; ++k)
                        if (k < B) {
                            const double x = sh[k * d.np + j];
#pragma unroll
                            for (int r = 0; r < 4; ++r) s4[k][r] += row[r] * x;
                        }
                }
#pragma unroll
                for (int k = 0; k < TRY_B; ++k)
                    if (k < B) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s4[k][r] = wave_sum(s4[k][r]);
                        if (lane == 0)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (i0 + r < d.np) {
                                    const double t = s4[k][r] + p.pinfo[i0 + r];
                                    red[2 * k] += 0.5 * t * t;
                                }
                    }
            }
        }
        KPROF(17);
        for (int e = tid; e < B * d.F; e += nt) {
            const int k = e / d.F, f = e - k * d.F;
            const double *a = p.state + 16 * f, *b = p.cand + (size_t)k * 16 * d.F + 16 * f;
            double acc = 0;
            if (pose_free(p.fix[f]))
                for (int q = 0; q < 7; ++q) acc += (a[q] - b[q]) * (a[q] - b[q]);
            if (motion_free(p.fix[f]))
                for (int q = 7; q < 16; ++q) acc += (a[q] - b[q]) * (a[q] - b[q]);
#pragma unroll
            for (int q = 0; q < TRY_B; ++q)
                if (q == k) red[2 * q + 1] += acc;
        }
        for (int e = tid; e < B * d.L; e += nt) {
            const int k = e / d.L, l = e - k * d.L;
            if (!p.lact[l]) continue;
            const double df = p.depth[l] - p.depth_cand[(size_t)k * d.L + l];
#pragma unroll
            for (int q = 0; q < TRY_B; ++q)
                if (q == k) red[2 * q + 1] += df * df;
        }
        if (B == 1) {
            double r1[2] = {red[0], red[1]};
            block_sum_n<2>(r1, scratch);
            red[0] = r1[0];
            red[1] = r1[1];
        } else {
            block_sum_n<2 * TRY_B>(red, scratch);
        }
        KPROF(18);
        // ---- replay the decisions in order (uniform across the workgroup)
        int accepted = -1;
#pragma unroll
        for (int k = 0; k < TRY_B; ++k) {
            if (k >= B || t.status != ST_RUNNING) continue;
#ifdef XRHIP_KPROF
            if (tid == 0) p.ctl->prof[19] += 1;   // trials
#endif
            if (trial_decide(t, k, model_cost_change[k], red[2 * k], red[2 * k + 1], step_norm[k])) accepted = k;
        }
        if (accepted >= 0) {
            const double *cs = p.cand + (size_t)accepted * 16 * d.F, *cd = p.depth_cand + (size_t)accepted * d.L;
            for (int e = tid; e < 16 * d.F; e += nt) p.state[e] = cs[e];
            for (int l = tid; l < d.L; l += nt) p.depth[l] = cd[l];
        }
        __syncthreads();   // candidates / deltas are rewritten by the next batch
        if (t.status == ST_RUNNING && wide_after_first) t.status = ST_NEED_TRIALS;
        B = TRY_B;
    }
    if (tid == 0) trial_store(c, t);
    __syncthreads();
    publish_block(d, p, t.status, seq, publish_always);
    return t.status;
}

// ------------------------------------------------------------------ fused launches
// A kernel on this device costs ~4-5 us of dispatch + drain however little it does, and the phases of one
// trust-region round are small; independent phases therefore share a launch (block ranges select the role), and
// consecutive single-workgroup phases run back to back in one kernel.

// all four factor families at once: [obs | rot | imu (one factor per block) | prior (1 block)]
__device__ __forceinline__ void d_lin_all(const BaDims &d, const BaPtrs &p, const Ext &cam, const Ext &imu, double sx, double sy, int bx) {
    extern __shared__ double sh[];   // np doubles (prior role)
    __shared__ double scr[IMU_SCR + IMU_XCH];   // lin_imu_block: raw residual + the two raw Jacobians | the pieces' exchange area
    __shared__ double scratch[8];
    const int nbo = (d.M + 255) / 256, nbr = (d.MR + 255) / 256, nbi = d.NI;
    int blk = bx;
    if (blk < nbo) {
        const int o = blk * 256 + threadIdx.x;
        if (o < d.M) lin_obs_item(d, p, cam, sx, sy, o);
        return;
    }
    blk -= nbo;
    if (blk < nbr) {
        const int o = blk * 256 + threadIdx.x;
        if (o < d.MR) lin_rot_item(d, p, cam, sx, sy, o);
        return;
    }
    blk -= nbr;
    if (blk < nbi) {
        lin_imu_block(d, p, imu, blk, scr);
        return;
    }
    blk -= nbi;
    if (d.NP == 0) {
        if (threadIdx.x == 0) p.pcost[0] = 0.0;
        return;
    }
    lin_prior_rows_block(d, p, blk, sh, scratch);
}
__global__ __launch_bounds__(256) void kb_lin_all(BaDims d, BaPtrs p, Ext cam, Ext imu, double sx, double sy) {
    d_lin_all(d, p, cam, imu, sx, sy, blockIdx.x);
}
// blocks: [obs | rot | imu (one factor each) | prior: 16 rows each (one block when there is no prior: it clears the cost)]
__host__ __device__ __forceinline__ int lin_all_blocks(int M, int MR, int NI, int np) {
    return (M + 255) / 256 + (MR + 255) / 256 + NI + (np ? (np + 15) / 16 : 1);
}

// per-landmark rows and per-frame-pair reprojection blocks: [Lp landmarks | F*F pairs x VIS_CH list chunks], one wavefront each
// (tried: 64 landmarks per block, one THREAD walking a landmark's observation list -- no butterflies, but ~8 dependent
// round trips to L2 per landmark instead of one: 21.7 -> 24.1 us, profiles/r02_ab_variants.md)
// (no landmark rows are needed when every landmark is constant: nla == 0)
__device__ __forceinline__ void d_landmark_vision(const BaDims &d, const BaPtrs &p, int bx) {
    __shared__ double red[VIS_RED];
    const int lp = d.lm_rows;
#ifdef XRHIP_KPROF_PRINT
    const long long t0 = wall_clock64();
#endif
    if (bx < lp) landmark_item(d, p, bx, threadIdx.x);
    else assemble_vision_item(d, p, (bx - lp) / VIS_CH, threadIdx.x, red, (bx - lp) % VIS_CH, VIS_CH);
#ifdef XRHIP_KPROF_PRINT
    if (threadIdx.x == 0 && d.M > 1500 && wall_clock64() - t0 > 800)
        printf("kb_landmark_vision block %d of %d+%d*%d (%s): %lld x10ns\n", bx, lp, d.F, d.F, bx < lp ? "landmark" : "pair", wall_clock64() - t0);
#endif
}
__global__ __launch_bounds__(64) void kb_landmark_vision(BaDims d, BaPtrs p) { d_landmark_vision(d, p, blockIdx.x); }

// Speculative linearisation (window solves).  While kb_trials_wide costs the first candidate of a round on its 32
// workgroups, the rest of the chip linearises the problem AT that candidate -- spec_state_block forms it (the same
// trial_begin / dogleg_point / Plus expressions kb_trials_wide uses, hence the same bits) into a second set of buffers,
// and kb_lin_all .. kb_schur_aux run on it in a second stream.  The first candidate of a round is accepted in three
// rounds out of four (tests/golden/ba_snapshots): the host then swaps the buffer sets and goes straight to the
// factorisation, ~90 us of dependent launches earlier.  Otherwise the second set is simply never looked at.
// The candidate is formed by the factorisation kernel itself, right after it has produced the step (spec_state_block at the
// end of kb_solve_try): it depends on the minimiser's scalars and on the state, both of which kb_trials_wide rewrites
// when it finishes -- nothing that runs BESIDE kb_trials_wide may read them.  The linearisation chain on the second stream
// only reads the candidate (state2 / depth2 / ctl2) and the problem's constant inputs.
__device__ __forceinline__ void spec_state_block(const BaDims &d, const BaPtrs &p, double *state2, double *depth2, BaCtl *ctl2, int mode) {
    const int tid = threadIdx.x, nt = blockDim.x, n = d.n;
    TrialScalars t;
    trial_load(p.ctl, t);
    trial_begin(t, mode == 3, mode == 1);
    const bool ok = t.status == ST_RUNNING;
    double ca = 0.0, cb = 0.0, sn = 0.0;
    if (ok) dogleg_point(t, t.radius, ca, cb, sn);
    for (int f = tid; f < d.F; f += nt) {
        double dl[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const int a = 15 * f + q;
            dl[q] = ((ca * p.grad[a] + cb * p.gn[a]) / p.diagD[a]) * p.sp[a];
        }
        state_plus(p.state + 16 * f, dl, pose_free(p.fix[f]), motion_free(p.fix[f]), state2 + 16 * f);
    }
    for (int l = tid; l < d.L; l += nt) {
        const double dep = p.depth[l];
        depth2[l] = p.lact[l] ? dep + ((ca * p.grad[n + l] + cb * p.gn[n + l]) / p.diagD[n + l]) * p.sl[l] : dep;
    }
    if (tid == 0) {   // the two fields the linearisation chain reads from its control block
        ctl2->mu = fmax(1e-8, 2.0 * t.mu / 10.0);   // what trial_decide leaves behind when it accepts a step
        ctl2->first = 0;
        ctl2->status = ok ? 1 : 0;
    }
}
// reduced-system solve followed by the trust-region trials, one workgroup
// wide_trials: 1 = a rejected first trial hands over to kb_trials_wide instead of looping in here; 2 = every trial,
// the first included, is costed by kb_trials_wide (queued behind this kernel by the host)
// NT = workgroup size.  The trial code holds whole IMU records and reprojection chains in registers: at 512 threads
// (256 VGPRs each) it spills ~190 registers to scratch memory, at 256 threads it gets 512 registers and spills
// nothing -- small problems (one observation per thread either way) run the 256-thread instance.
// PREP = true is the wide_trials == 2 instance: the trial code is not even compiled in, so the factorisation of a
// window-sized system runs in a 512-thread kernel without the ~190 spilled registers the trial loop costs there.
// state2 / depth2 / ctl2 (PREP only, may be null): where the first candidate goes for the speculative linearisation
template <int NT, bool PREP>
__device__ __forceinline__ void d_solve_try(const BaDims &d, const BaPtrs &p, const Ext &cam, const Ext &imu, double sx, double sy, int use_lds,
                                            int after_linearisation, int seq, int wide_trials, double *state2, double *depth2,
                                            BaCtl *ctl2, int commit) {
    extern __shared__ double lds[];   // max(solve_block's region, try_block's staging)
    if (PREP && commit) {   // the speculation was right: the cost and gradient norm of its linearisation become the minimiser's
        if (threadIdx.x == 0) {
            p.ctl->x_cost = ctl2->x_cost;
            p.ctl->gmax = ctl2->gmax;
        }
        __syncthreads();
    }
    solve_block(d, p, use_lds, lds);
    __syncthreads();
    if (PREP) {
        try_block(d, p, cam, imu, sx, sy, after_linearisation, seq, true, lds, true);
        if (state2) {
            __syncthreads();   // thread 0's control-block stores of the phase above
            spec_state_block(d, p, state2, depth2, ctl2, after_linearisation);
        }
    } else {
        try_block(d, p, cam, imu, sx, sy, after_linearisation, seq, wide_trials != 0, lds, false);
    }
}
template <int NT, bool PREP>
__global__ __launch_bounds__(NT) void kb_solve_try(BaDims d, BaPtrs p, Ext cam, Ext imu, double sx, double sy, int use_lds,
                                                   int after_linearisation, int seq, int wide_trials, double *state2, double *depth2,
                                                   BaCtl *ctl2, int commit) {
    d_solve_try<NT, PREP>(d, p, cam, imu, sx, sy, use_lds, after_linearisation, seq, wide_trials, state2, depth2, ctl2, commit);
}

// The rejected-trial tail of a large solve on the whole chip.  After a rejection the next radii are radius/2,
// radius/4, ...: WIDE_B candidates are costed by WIDE_G workgroups, each taking a slice of the factors (the small
// candidate states are recomputed per workgroup in LDS); per-block partial sums go to global memory, and the last
// block to arrive adds them up in block order, replays the accept / reject decisions exactly like try_block,
// applies an accepted step and publishes to the host mailbox.  ST_NEED_TRIALS = all WIDE_B rejected, launch again.
// Dynamic LDS: WIDE_B * (16 F + np) doubles.
// first != 0: this launch carries the first trial after a solve (mode = the solve's after_linearisation flag), so it
// starts with the minimiser's finalize / start-of-iteration step instead of continuing a run of rejections.
__device__ __forceinline__ void d_trials_wide(const BaDims &d, const BaPtrs &p, const Ext &cam, const Ext &imu, double sx, double sy, int seq,
                                              int first, int mode, int bx, int gx) {
    extern __shared__ double wl[];
    __shared__ double scratch[4 * WIDE_B * 4];
    __shared__ int s_last;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int blk = bx, G = gx;
    const int n = d.n;
#ifdef XRHIP_KPROF_PRINT
    long long tw[16];
    int twn = 0;
#define WT() (tw[twn++] = wall_clock64())
#else
#define WT() ((void)0)
#endif
    WT();
    double *cand = wl;                           // [WIDE_B][F][16]
    double *pd = wl + (size_t)WIDE_B * 16 * d.F;   // [WIDE_B][np]
    BaCtl *c = p.ctl;
    TrialScalars t;
    trial_load(c, t);
    if (first) {
        trial_begin(t, mode == 3, mode == 1);
        if (t.status != ST_RUNNING) {   // terminated / re-solve requested before any trial: every block agrees
            if (blk == 0) {
                if (tid == 0) trial_store(c, t);
                __syncthreads();
                publish_block(d, p, t.status, seq, true);
            }
            return;
        }
    }
    // otherwise the batch continues a run of rejections: trial 0 replays the finalize step of its rejected predecessor
    // Slot -> trial.  While the Gauss-Newton step lies inside the radius the dogleg point IS that step, whatever the
    // radius: the first `dup` trials of the batch (radius, radius / 2, ... all still >= |gn|) share one candidate and one
    // cost -- slot 0 stands for all of them and the other slots continue behind them.  (The reference's live-bias quirk
    // ends most refine_window solves in ~25 rejections, about half of them of this kind: the S1 snapshot of
    // tests/golden/ba_snapshots repeats one candidate 12 times.)
    int dup = 0;
    {
        double rk = t.radius;
        while (dup < 60 && t.gn_norm <= rk) {
            ++dup;
            rk *= 0.5;
        }
        if (dup < 1) dup = 1;
    }
    double ca[WIDE_B], cb[WIDE_B], step_norm[WIDE_B];
#pragma unroll
    for (int k = 0; k < WIDE_B; ++k) {
        const int j = k == 0 ? 0 : dup + k - 1;   // trial index of slot k
        dogleg_point(t, scalbn(t.radius, -j), ca[k], cb[k], step_norm[k]);
    }
    WT();
    // ---- candidate frame states (every block, LDS) and prior deltas
    for (int e = tid; e < WIDE_B * d.F; e += nt) {
        const int k = e / d.F, f = e - k * d.F;
        double dl[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const int a = 15 * f + q;
            double cak = 0, cbk = 0;
#pragma unroll
            for (int kk = 0; kk < WIDE_B; ++kk)
                if (kk == k) {
                    cak = ca[kk];
                    cbk = cb[kk];
                }
            dl[q] = ((cak * p.grad[a] + cbk * p.gn[a]) / p.diagD[a]) * p.sp[a];
        }
        state_plus(p.state + 16 * f, dl, pose_free(p.fix[f]), motion_free(p.fix[f]), cand + (size_t)(k * d.F + f) * 16);
    }
    __syncthreads();
    WT();
    for (int e = tid; e < WIDE_B * d.NP; e += nt) {
        const int k = e / d.NP, i = e - k * d.NP;
        double dl[15];
        prior_delta(p, i, cand + (size_t)k * 16 * d.F, dl, nullptr);
        for (int q = 0; q < 15; ++q) pd[k * d.np + 15 * i + q] = dl[q];
    }
    __syncthreads();
    WT();
    // ---- this block's slice of the sums: per candidate cost, |x - cand|^2, |step|^2, step . gs
    double acc[4 * WIDE_B];
#pragma unroll
    for (int q = 0; q < 4 * WIDE_B; ++q) acc[q] = 0.0;
    // With enough wavefronts in the grid the (IMU factor, candidate) pairs -- one wavefront each, a serial SO(3) chain of 5-9 us -- get
    // wavefronts of their own (the far end of the grid) and the other wavefronts share the strided sums: the chain then runs beside the
    // reprojection factors and the prior rows instead of behind them on the same wavefront (in-kernel timers, round 3).
    const int W = G * nw, T = WIDE_B * d.NI, gw = blk * nw + wave;
    const bool split = 2 * T <= W;
    const bool strided = !(split && gw >= W - T);
    const int gtid = blk * nt + tid, gnt = split ? (W - T) * 64 : G * nt, wstride = split ? W - T : W;
    // (these few items go to the far end of the strided range: its near end is where a second reprojection pair per thread lands)
    if (strided)
    for (int a = gnt - 1 - gtid; a < d.NV; a += gnt) {   // step norms / gradient products, landmark part of |x - cand|^2
        const double g = p.grad[a], gnv = p.gn[a], D = p.diagD[a], gsa = p.gs[a];
        const bool lm = a >= n;
        const double sl = lm ? p.sl[a - n] : 0.0;
        const bool act = lm && p.lact[a - n];
#pragma unroll
        for (int k = 0; k < WIDE_B; ++k) {
            const double v = ca[k] * g + cb[k] * gnv;
            acc[4 * k + 2] += v * v;
            const double st = v / D;
            acc[4 * k + 3] += st * gsa;
            if (act) {   // |depth - candidate depth|^2 from the stored values, like try_block
                const double dep = p.depth[a - n];
                const double df = dep - (dep + st * sl);
                acc[4 * k + 1] += df * df;
            }
        }
    }
    if (strided)
        for (int e = gnt - 1 - gtid - d.NV; e >= 0 && e < WIDE_B * d.F; e += gnt) {
            const int k = e / d.F, f = e - k * d.F;
            const double *a = p.state + 16 * f, *b = cand + (size_t)(k * d.F + f) * 16;
            double s2 = 0;
            if (pose_free(p.fix[f]))
                for (int q = 0; q < 7; ++q) s2 += (a[q] - b[q]) * (a[q] - b[q]);
            if (motion_free(p.fix[f]))
                for (int q = 7; q < 16; ++q) s2 += (a[q] - b[q]) * (a[q] - b[q]);
#pragma unroll
            for (int kk = 0; kk < WIDE_B; ++kk)
                if (kk == k) acc[4 * kk + 1] += s2;
        }
    WT();
    if (strided)
    for (int e = gtid; e < WIDE_B * d.M; e += gnt) {   // reprojection factors: one (factor, candidate) pair per thread
        const int k = e / d.M, o = e - k * d.M;
        const int l = p.obs_lm[o];
        double cak = 0, cbk = 0;
#pragma unroll
        for (int kk = 0; kk < WIDE_B; ++kk)
            if (kk == k) {
                cak = ca[kk];
                cbk = cb[kk];
            }
        const double dep = p.depth[l];
        const double dk = p.lact[l] ? dep + ((cak * p.grad[n + l] + cbk * p.gn[n + l]) / p.diagD[n + l]) * p.sl[l] : dep;
        const double cst = obs_cost_at(d, p, o, cand + (size_t)k * 16 * d.F, dk, cam, sx, sy);
#pragma unroll
        for (int kk = 0; kk < WIDE_B; ++kk)
            if (kk == k) acc[4 * kk] += cst;
    }
    if (strided)
    for (int e = gtid; e < WIDE_B * d.MR; e += gnt) {
        const int k = e / d.MR, o = e - k * d.MR;
        const double cst = rot_eval(d, p, o, cand + (size_t)k * 16 * d.F, cam, sx, sy, false, nullptr);
#pragma unroll
        for (int kk = 0; kk < WIDE_B; ++kk)
            if (kk == k) acc[4 * kk] += cst;
    }
    // IMU factors: raw residual + 15x15 whitening per (factor, candidate), taken by the blocks from the far end of the
    WT();
    // grid so that they do not pile onto the threads that already hold a reprojection pair
    // (one WAVEFRONT per (factor, candidate) pair: lane 0 runs the SO(3) chain of the raw residual, fifteen lanes whiten
    // it -- as a single thread the 225-term whitening doubled the chain, and these pairs are the last to finish)
    for (int e = W - 1 - gw; e < T; e += W) {
        const int k = e / d.NI, f = e - k * d.NI;
        const double cst = imu_cost_wave(p, f, cand + (size_t)k * 16 * d.F, imu, lane);   // in lane 0, zero elsewhere
#pragma unroll
        for (int kk = 0; kk < WIDE_B; ++kk)
            if (kk == k) acc[4 * kk] += cst;
    }
    WT();
    if (strided)
    for (int i = gw; i < d.np; i += wstride) {   // prior rows: one wavefront per row, every candidate
        double sr[WIDE_B];
#pragma unroll
        for (int k = 0; k < WIDE_B; ++k) sr[k] = 0.0;
        for (int j = lane; j < d.np; j += 64) {
            const double sij = p.pS[(size_t)i * d.np + j];
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) sr[k] += sij * pd[k * d.np + j];
        }
#pragma unroll
        for (int k = 0; k < WIDE_B; ++k) {
            const double r = wave_sum(sr[k]) + p.pinfo[i];
            if (lane == 0) acc[4 * k] += 0.5 * r * r;
        }
    }
    WT();
    // ---- block partial sums -> global; the last block to finish reduces and decides.  The 32 sums of a block go through
    // an LDS tile [32][257] (every thread parks its terms as a column; 128 threads add up a quarter row each, 32 combine the
    // quarters): 32 butterfly reductions of doubles -- 12 ds_bpermute each -- took a fifth of this kernel.
    {
        double *tile = wl + (size_t)WIDE_B * (16 * d.F + d.np);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4 * WIDE_B; ++q) tile[q * 257 + tid] = acc[q];
        __syncthreads();
        if (tid < 4 * 4 * WIDE_B) {
            const int q = tid >> 2, w4 = tid & 3;
            const double *row = tile + q * 257 + 64 * w4;
            double s2 = 0.0;
            for (int i = 0; i < 64; ++i) s2 += row[i];
            scratch[tid] = s2;
        }
        __syncthreads();
        if (tid < 4 * WIDE_B)
            p.wide_part[(size_t)blk * 4 * WIDE_B + tid] = ((scratch[4 * tid] + scratch[4 * tid + 1]) + scratch[4 * tid + 2]) + scratch[4 * tid + 3];
    }
    __syncthreads();
    WT();
    if (tid == 0) {
        __threadfence();
        const unsigned ticket = atomicAdd(&c->wide_ticket, 1u);
        s_last = (ticket == (unsigned)G - 1u) ? 1 : 0;
    }
    __syncthreads();
#ifdef XRHIP_KPROF_PRINT
    if (!s_last && tid == 0 && d.M > 1500 && (blk == 0 || blk == G - 1 || blk == G / 2))
        printf("kb_trials_wide block %d: load+dogleg %lld cand %lld prior_delta %lld nv %lld obs+rot %lld imu %lld prior %lld reduce %lld (x10ns)\n", blk, tw[1] - tw[0],
               tw[2] - tw[1], tw[3] - tw[2], tw[4] - tw[3], tw[5] - tw[4], tw[6] - tw[5], tw[7] - tw[6], tw[8] - tw[7]);
#endif
    if (!s_last) return;
    WT();
    // The other blocks' partial sums are read past the caches (device-scope loads) instead of behind a device-scope acquire fence:
    // that fence invalidates this XCD's L2 and, with the loads behind it, took 7-10 us of the last block (in-kernel timers, round 3).
    // The ticket orders them: every block's sums are written back (its __threadfence) before its ticket, and these loads are issued
    // after this block has seen the last ticket.
    static_assert(WIDE_G % 4 == 0, "the last block adds the partial sums up in four runs of WIDE_G / 4 blocks");
    __shared__ double quarter[4 * 4 * WIDE_B];
    if (tid < 4 * 4 * WIDE_B) {   // four threads per sum, a quarter of the blocks each (block order), combined in quarter order below
        const int q = tid >> 2, w4 = tid & 3;
        const double *part = static_cast<const double *>(p.wide_part) + q;
        double v[WIDE_G / 4];
#pragma unroll
        for (int i = 0; i < WIDE_G / 4; ++i) {
            const int b = w4 * (WIDE_G / 4) + i;
            v[i] = b < G ? __hip_atomic_load(part + (size_t)b * 4 * WIDE_B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
        double s2 = 0;
#pragma unroll
        for (int i = 0; i < WIDE_G / 4; ++i)
            if (w4 * (WIDE_G / 4) + i < G) s2 += v[i];
        quarter[tid] = s2;
    }
    __syncthreads();
    if (tid < 4 * WIDE_B) scratch[tid] = ((quarter[4 * tid] + quarter[4 * tid + 1]) + quarter[4 * tid + 2]) + quarter[4 * tid + 3];
    __syncthreads();
    WT();
    double tot[4 * WIDE_B];
#pragma unroll
    for (int q = 0; q < 4 * WIDE_B; ++q) tot[q] = scratch[q];
    double mcc[WIDE_B];
    TrialTerms terms[WIDE_B];
#pragma unroll
    for (int k = 0; k < WIDE_B; ++k) {
        if (step_norm[k] < 0) step_norm[k] = sqrt(tot[4 * k + 2]);
        mcc[k] = dogleg_model_change(t, ca[k], cb[k], tot[4 * k + 3]);
        terms[k] = trial_terms(t, mcc[k], tot[4 * k], tot[4 * k + 1]);
    }
    int accepted = -1;
#ifdef XRHIP_KPROF
    int decisions = 0;
#endif
    // the decisions, trial by trial: `dup` of them on slot 0's sums, then one per remaining slot.  As the continuation of a
    // rejection run every trial of the batch, the first included, starts with the finalize step of its rejected predecessor
    // (trial index j + 1 > 0).  (Written as one loop over j with the slot's terms picked by a select chain, the 15-25 decisions
    // of a launch cost 7 us of issue slots: ~100 instructions each at one wavefront's ~8 cycles per instruction.)
    int j = 0;
    for (; j < dup && t.status == ST_RUNNING; ++j) {
        if (trial_decide(t, first ? j : j + 1, mcc[0], terms[0], step_norm[0])) accepted = 0;
#ifdef XRHIP_KPROF
        ++decisions;
#endif
    }
#pragma unroll
    for (int k = 1; k < WIDE_B; ++k) {
        if (t.status != ST_RUNNING) continue;
        const int jj = dup + k - 1;
        if (trial_decide(t, first ? jj : jj + 1, mcc[k], terms[k], step_norm[k])) accepted = k;
#ifdef XRHIP_KPROF
        ++decisions;
#endif
    }
#ifdef XRHIP_KPROF
    if (tid == 0) p.ctl->prof[19] += decisions;
#endif
    WT();
    if (accepted >= 0) {
        for (int e = tid; e < 16 * d.F; e += nt) p.state[e] = cand[(size_t)accepted * 16 * d.F + e];
        double cak = 0, cbk = 0;
#pragma unroll
        for (int kk = 0; kk < WIDE_B; ++kk)
            if (kk == accepted) {
                cak = ca[kk];
                cbk = cb[kk];
            }
        for (int l = tid; l < d.L; l += nt)
            if (p.lact[l]) p.depth[l] = p.depth[l] + ((cak * p.grad[n + l] + cbk * p.gn[n + l]) / p.diagD[n + l]) * p.sl[l];
    }
    if (t.status == ST_RUNNING) t.status = ST_NEED_TRIALS;
    if (tid == 0) {
        trial_store(c, t);
        c->wide_ticket = 0;
        c->accepted_slot = accepted;
    }
    __syncthreads();
    WT();
    publish_block(d, p, t.status, seq, true);
#ifdef XRHIP_KPROF_PRINT
    if (tid == 0 && d.M > 1500)
        printf("kb_trials_wide LAST block %d: load+dogleg %lld cand %lld prior_delta %lld nv %lld obs+rot %lld imu %lld prior %lld reduce %lld ticket %lld fence+sum %lld decide %lld apply %lld publish %lld (x10ns)\n", blk,
               tw[1] - tw[0], tw[2] - tw[1], tw[3] - tw[2], tw[4] - tw[3], tw[5] - tw[4], tw[6] - tw[5], tw[7] - tw[6], tw[8] - tw[7], tw[9] - tw[8],
               tw[10] - tw[9], tw[11] - tw[10], tw[12] - tw[11], wall_clock64() - tw[12]);
#endif
#undef WT
}
__global__ __launch_bounds__(256) void kb_trials_wide(BaDims d, BaPtrs p, Ext cam, Ext imu, double sx, double sy, int seq,
                                                      int first, int mode) {
    d_trials_wide(d, p, cam, imu, sx, sy, seq, first, mode, blockIdx.x, gridDim.x);
}

// Small problems without a free landmark (localize_newframe, refine_subwindow: a handful of free dofs): the three
// launches between the frame-pair blocks and the solve -- Hessian assembly, cost / gradient norm / preparation, the
// reduced system + Cauchy quadratic form -- are one workgroup's worth of work and run back to back here.
// Only what the solve reads is assembled: the active x active entries (+ the gradient), zero diagonals elsewhere.
__device__ __forceinline__ void small_mid_block(const BaDims &d, const BaPtrs &p, int relinearised, double *scratch) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int n = d.n, na = d.na;
    if (relinearised) {
        const bool col0_active = dof_active(p.fix, 0);
        for (int e = tid; e < na * (na + 1); e += nt) {
            const int i = e / (na + 1), j = e - i * (na + 1);
            if (j == na && col0_active) continue;   // the gradient entry rides on column 0, already visited
            assemble_item(d, p, p.act_idx[i] * n + (j < na ? p.act_idx[j] : 0));
        }
        for (int a = tid; a < n; a += nt)
            if (!dof_active(p.fix, a)) {
                p.Hpp[(size_t)a * n + a] = 0.0;
                p.gp[a] = 0.0;
            }
        __syncthreads();
        sum_cost_block(d, p, scratch);
        __syncthreads();
        gradmax_block(d, p, scratch);
    }
    prepare_block(d, p);
    __syncthreads();
    for (int blk = 0; blk < sred_rest_blocks(d); ++blk) reduced_rest_block(d, p, blk);
    double acc = 0;   // Q(g~,g~) over the active dofs (see kb_schur_aux)
    for (int i = wave; i < na; i += nw) {
        const int a = p.act_idx[i];
        const double Da = p.diagD[a], ga = p.sp[a] * (p.gs[a] / (Da * Da));
        double t = 0;
        for (int j = lane; j < na; j += 64) {
            const int b = p.act_idx[j];
            const double Db = p.diagD[b];
            t += p.Hpp[(size_t)a * n + b] * (p.sp[b] * (p.gs[b] / (Db * Db)));
        }
        acc += ga * t;
    }
    acc = block_sum(acc, scratch);
    const int nbq = aux_quad_blocks_n(n, d.L);
    for (int i = tid; i < nbq; i += nt) p.partial[i] = (i == 0) ? acc : 0.0;
}

__global__ __launch_bounds__(256) void kb_small_mid(BaDims d, BaPtrs p, int relinearised) {
    __shared__ double scratch[8];
    small_mid_block(d, p, relinearised, scratch);
}

// A whole solve in ONE launch for problems without free landmarks (localize_newframe, refine_subwindow, the
// initialiser's PnP): a handful of free frames, <= a few hundred constant-landmark observations, the IMU factors
// between them.  One workgroup runs the trust-region loop itself -- linearise, assemble the active block, prepare,
// factor, trials -- with the same device bodies the multi-launch path uses, and publishes once, when the minimiser
// terminates.  Saves 5 launches and one host round trip per round.
// Dynamic LDS: max(solve_block's region, try_block's staging, np doubles for the prior).
// The argument block lives in device memory (it is staged with the problem): the loop below keeps very little of it
// in registers, where by-value kernel arguments would pin ~250 scalar registers across every phase.
struct TinyArgs {
    BaDims d;
    BaPtrs p;
    Ext cam, imu;
    double sx, sy;
};

// ---------------------------------------------------------------------------------------------- window rounds of an instance group
// Round 5.  A keyframe's refine_window is ~45 launches; the members of an instance group used to issue them one by one on their own
// streams, which share two hardware queues with the other members' rounds and marginalisations (2.1 ms per keyframe against 1.05
// solo; 3.5 of 8 members are inside such rounds at any moment: profiles/r05_multi_sequence.md).  Here the kernels of a round take
// several members' rounds at once: blockIdx.z = entry, every entry with its own argument block in device memory (the TinyArgs staged
// with the problem: BaDims + BaPtrs are ~0.7 KB, eight of them do not fit the 4 KB a launch can carry by value) and its own grid sizes
// -- blocks beyond an entry's grid return at once.  The bodies are the solo kernels' (d_lin_all ... d_trials_wide): same block-to-work
// mapping, same summation order, same bits.
struct WinEntry {
    const TinyArgs *args;
    int g_lin, g_lv, g_asm, g_sa;    // this entry's grid sizes; 0: the kernel is not part of the entry's round (no relinearisation)
    int use_lds, mode, seq, first;   // kb_solve_try's layout switch and after_linearisation flag, the mailbox sequence number; kw_trials_wide: first
};
__global__ __launch_bounds__(256) void kw_lin_all(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if ((int)blockIdx.x >= e.g_lin) return;
    const TinyArgs &a = *e.args;
    d_lin_all(a.d, a.p, a.cam, a.imu, a.sx, a.sy, blockIdx.x);
}
__global__ __launch_bounds__(64) void kw_landmark_vision(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if ((int)blockIdx.x >= e.g_lv) return;
    d_landmark_vision(e.args->d, e.args->p, blockIdx.x);
}
__global__ __launch_bounds__(256) void kw_assemble(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if ((int)blockIdx.x >= e.g_asm) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < e.args->d.n * e.args->d.n) assemble_item(e.args->d, e.args->p, idx);
}
__global__ __launch_bounds__(256) void kw_prepare(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if (!e.args) return;
    prepare_block(e.args->d, e.args->p);
}
__global__ __launch_bounds__(256) void kw_schur_aux(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if ((int)blockIdx.x >= e.g_sa) return;
    d_schur_aux(e.args->d, e.args->p, blockIdx.x, e.g_sa);
}
__global__ __launch_bounds__(512) void kw_solve_try(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if (!e.args) return;
    const TinyArgs &a = *e.args;
    d_solve_try<512, true>(a.d, a.p, a.cam, a.imu, a.sx, a.sy, e.use_lds, e.mode, e.seq, 2, nullptr, nullptr, nullptr, 0);
}
__global__ __launch_bounds__(256) void kw_trials_wide(Batch<WinEntry> b) {
    const WinEntry &e = b.e[blockIdx.z];
    if (!e.args) return;
    const TinyArgs &a = *e.args;
    d_trials_wide(a.d, a.p, a.cam, a.imu, a.sx, a.sy, e.seq, e.first, e.mode, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void kb_tiny(const TinyArgs *__restrict__ args, int use_lds, int seq, int max_rounds) {
    const BaDims &d = args->d;
    const BaPtrs &p = args->p;
    const Ext &cam = args->cam, &imu = args->imu;
    const double sx = args->sx, sy = args->sy;
    extern __shared__ double lds[];
    __shared__ double scr[4][IMU_SCR];
    __shared__ double scratch[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_next;
    __shared__ double s_vis[4][27];
    __shared__ int s_free[64], s_nfree;   // frames with a free pose: only their pairs carry a reprojection block
    __shared__ double s_red[VIS_RED];
    if (tid == 0) {
        int nf = 0;
        for (int f = 0; f < d.F; ++f)
            if (pose_free(p.fix[f]) && nf < 64) s_free[nf++] = f;
        s_nfree = nf;
    }
    __syncthreads();
    bool relin = true;
    int mode = 1, st = ST_RUNNING;
    for (int round = 0; round < max_rounds; ++round) {
        KPROF_BEGIN();
        if (relin) {
            // the IMU factors (a long serial chain on one lane each) take the first wavefronts; the others work
            // through the observations meanwhile, 64 at a time from a shared counter, and are joined by the IMU
            // wavefronts as those finish
            if (tid == 0) s_next = 0;
            __syncthreads();
            for (int k = wave; k < d.NI; k += 4) lin_imu_item(d, p, imu, k, lane, scr[wave]);
            for (;;) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_next, 64);
                base = __shfl(base, 0);
                if (base >= d.M + d.MR) break;
                const int o = base + lane;
                if (o < d.M) lin_obs_item(d, p, cam, sx, sy, o);
                else if (o < d.M + d.MR) lin_rot_item(d, p, cam, sx, sy, o - d.M);
            }
            KPROF(20);
            lin_prior_block(d, p, lds, scratch);
            __syncthreads();
            KPROF(21);
            if (s_nfree == 1) {
                // the one 6x6 reprojection block (f, f): its observations are shared out over all four wavefronts
                // (assemble_vision_item would leave three of them idle); upper triangle + gradient = 27 sums
                const int f = s_free[0], pair = f * d.F + f;
                const int s0 = p.pair_start[pair], s1 = p.pair_start[pair + 1];
                double acc[27];
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] = 0.0;
                for (int it = s0 + tid; it < s1; it += 256) {
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
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] = wave_sum(acc[i]);
                if (lane == 0)
#pragma unroll
                    for (int i = 0; i < 27; ++i) s_vis[wave][i] = acc[i];
                __syncthreads();
                if (tid < 36) {
                    const int a = tid / 6, b = tid - 6 * a, lo = a < b ? a : b, hi = a < b ? b : a;
                    const int e = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);   // index in the row-major upper triangle
                    p.Hv[(size_t)pair * 36 + tid] = (s_vis[0][e] + s_vis[1][e]) + (s_vis[2][e] + s_vis[3][e]);
                    for (int c = 1; c < VIS_CH; ++c) p.Hv[((size_t)c * d.F * d.F + pair) * 36 + tid] = 0.0;   // (vis_h adds the chunks)
                } else if (tid < 42) {
                    const int a = tid - 36;
                    p.gv[6 * f + a] = (s_vis[0][21 + a] + s_vis[1][21 + a]) + (s_vis[2][21 + a] + s_vis[3][21 + a]);
                    for (int c = 1; c < VIS_CH; ++c) p.gv[(size_t)c * 6 * d.F + 6 * f + a] = 0.0;
                }
            } else if (wave == 0) {   // (a path kb_chain has taken over for every problem it accepts: kept simple)
                for (int it = 0; it < s_nfree * s_nfree; ++it)
                    assemble_vision_item(d, p, s_free[it / s_nfree] * d.F + s_free[it % s_nfree], lane, s_red);
            }
            __syncthreads();
            KPROF(22);
        }
        small_mid_block(d, p, relin ? 1 : 0, scratch);
        __syncthreads();
        KPROF(23);
        solve_block(d, p, use_lds, lds);
        __syncthreads();
        KPROF(25);
        st = try_block(d, p, cam, imu, sx, sy, mode, seq, false, lds, false, false);
        KPROF(26);
#ifdef XRHIP_KPROF
        if (tid == 0) p.ctl->prof[27] += 1;   // rounds
#endif
        if (st == ST_DONE) return;
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
    // not terminated within the round budget (or an unexpected status): tell the host, which reports an error
    if (tid == 0) {
        p.ctl->status = st == ST_DONE ? ST_DONE : -1;
        const long long *src = reinterpret_cast<const long long *>(static_cast<BaCtl *>(p.ctl));
        long long *dst = reinterpret_cast<long long *>(static_cast<BaCtl *>(p.host_ctl));
        for (unsigned i = 0; i < sizeof(BaCtl) / sizeof(long long); ++i) dst[i] = src[i];
        __threadfence_system();
        *reinterpret_cast<volatile int *>(static_cast<int *>(p.host_seq)) = seq;
    }
}

// The staged problem (a few tens of KB) is pulled from the pinned host arena by the device itself, 16 bytes per
// lane: a kernel in the solve's own stream starts within a few microseconds, where a copy-engine transfer adds
// its scheduling latency in front of the first linearisation.
struct StageArgs {
    const uint4 *src_host;
    uint4 *dst;
    size_t n16;
};
__global__ __launch_bounds__(256) void kb_stage(Batch<StageArgs> b) {   // blockIdx.z = entry (one staged problem each)
    const StageArgs &a = b.e[blockIdx.z];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n16; i += (size_t)gridDim.x * 256) a.dst[i] = a.src_host[i];
}

}   // namespace xrhip
