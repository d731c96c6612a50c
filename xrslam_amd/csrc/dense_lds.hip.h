// dense_lds.hip.h -- workgroup-cooperative dense SPD kernels on a packed lower-triangular matrix that
// lives in LDS (or, for systems that exceed the 160 KB LDS of a CU, in an L2-resident global buffer).
//
// Used by the reduced-camera-system solve of the bundle adjustment (solve_block) and by the Cholesky
// fast path of the marginalisation (km_chol).  Right-looking blocked Cholesky with 16-wide panels:
//   (1) the 16x16 diagonal block is factored AND inverted by wavefront 0 entirely in registers: lane l owns
//       row l, values cross lanes through v_readlane -- no LDS round trip, no barrier on the serial chain;
//   (2) the panel below it is multiplied by the inverse on the f64 matrix cores (16-row tiles);
//   (3) the trailing matrix gets its rank-16 update from the matrix cores as well (16x16 tiles),
// i.e. 3 workgroup barriers per panel.  A right-hand side can ride along as an extra row of the matrix
// (nrows = n + 1): when the factorisation ends that row holds L^-1 rhs, so no forward solve is needed.
// The triangular solves run the 16x16 diagonal systems in registers the same way.
#pragma once
#include <hip/hip_runtime.h>

namespace xrhip {

constexpr int CH_NB = 16;

__host__ __device__ __forceinline__ int tri_idx(int i, int j) { return i * (i + 1) / 2 + j; }   // j <= i

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wavefront execute in program order; this only stops the compiler from
    // moving accesses across the point where other lanes' data is consumed.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// value of `v` in lane `src` (wave-uniform index), broadcast to every lane
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

#ifdef XRHIP_KPROF
#define CHPROF(slot)                                  \
    do {                                              \
        if (prof && threadIdx.x == 0) {               \
            const long long t_now = wall_clock64();   \
            prof[slot] += t_now - t_prev;             \
            t_prev = t_now;                           \
        }                                             \
    } while (0)
#else
#define CHPROF(slot) \
    do {             \
    } while (0)
#endif

// ---- The 16x16 diagonal block (round 6).
//
// What a single wavefront's double-precision code costs on gfx950 (tools/issue.hip, profiles/r06_issue.json): ~9 cycles per f64
// operation WHETHER OR NOT it depends on the one before (the "32-36 cycles of dependent latency" of rounds 2-5 was the loop branch
// of tools/latency.hip's un-unrolled chains), ~6-10 per 64-bit v_readlane broadcast, 20 per v_rcp_f64 / v_rsq_f64, 80 per
// v_mfma_f64_16x16x4 (dependent or not), ~80 per LDS round trip.  The block is therefore bound by the NUMBER of operations its one
// wavefront issues, and rounds 1-5's column-by-column form issued ~1000 of them: per pivot a reciprocal square root with two coupled
// Newton steps, a column scaling, and per later column a broadcast, a multiply and a subtract.
//
// Now:
//   * 2x2 block pivots.  For the pivot block [a b; b d] of columns (c, c+1) the Schur update of every later column k is
//         X[k] += xk0 U + xk1 V,   U = -(d x0 - b x1) / det,   V = -(a x1 - b x0) / det,   det = a d - b^2
//     (x0, x1: this lane's entries in the two pivot columns; xk0, xk1: row k's, broadcast): two fused operations and two
//     broadcasts per column and PAIR of pivots, one reciprocal (v_rcp_f64 + a cubic correction) per pair, no square root.
//   * The Cholesky factor's columns are the unscaled ones times a reciprocal square root: L[i][c] = x0 / sqrt(a),
//     L[i][c+1] = v / sqrt(a det), v = a x1 - b x0.  The sixteen arguments (a in lane c -- it is that lane's own x0 --, a det in
//     lane c+1 -- a times that lane's own v) are collected in ONE register across the lanes and go through v_rsq_f64 + correction
//     once, at the end; the positivity test of all pivots is one comparison of that register.
//   * Rows that only ride along (unit rows: they come out as the columns of L^-1; a right-hand side; rows of the panel below) sit
//     in the other lanes and get the same column operations for free.
// Same mathematics as the column-by-column form, other roundings (the 2x2 inverse is explicit): factors agree to a few ulp
// (tools/chol_test.hip prints both against a host Cholesky).

// 1 / x: v_rcp_f64 (24 bits, profiles/r06_issue.json) and the cubic step r0 (1 + e + e^2), e = 1 - x r0: 1.1e-16 relative
__device__ __forceinline__ double rcp_cubic(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    const double p = fma(e, e, e);
    return fma(r0, p, r0);
}
// 1 / sqrt(x): v_rsq_f64 (24 bits) and r0 (1 + e/2 + 3 e^2 / 8), e = 1 - x r0^2: 1.4e-16 relative
__device__ __forceinline__ double rsqrt_cubic(double x) {
    const double r0 = __builtin_amdgcn_rsq(x);
    const double xr = x * r0;
    const double e = fma(-xr, r0, 1.0);
    const double p = fma(0.375 * e, e, 0.5 * e);
    return fma(r0, p, r0);
}

// One wavefront.  X[k]: entry k of this lane's row.  Lanes 0-15 hold rows 0-15 of the block (lower triangle; what they hold above
// the diagonal is never read by another lane and may be anything finite), every other lane a row that rides along.  Columns >= nb
// are not pivots (padding, or a right-hand-side row that sits among the block's rows): they are left as they are.  On return X holds
// the lane's row of L (block rows) or row * L^-T (riding rows).  Returns false (wave-uniform) if a pivot block is not positive
// definite; X is then meaningless.
__device__ __forceinline__ bool diag16_pivot_pairs(double (&X)[CH_NB], int nb, int lane) {
    const int l16 = lane & 15;
    double S = 1.0;   // lane c (< 16): the number whose reciprocal square root scales column c
#pragma unroll
    for (int c = 0; c < CH_NB; c += 2) {
        const bool pv = c < nb, pw = c + 1 < nb;   // is column c / c + 1 a pivot
        double a = lane_bcast(X[c], c), b = lane_bcast(X[c], c + 1), d = lane_bcast(X[c + 1], c + 1);
        a = pv ? a : 1.0;
        b = pw ? b : 0.0;
        d = pw ? d : 1.0;
        const double x0 = X[c], x1 = X[c + 1];
        const double u = fma(d, x0, -(b * x1)), v = fma(a, x1, -(b * x0));   // in lane c + 1: v = a d - b^2 = det
        if (c + 2 < CH_NB) {
            const double det = fma(a, d, -(b * b));
            const double rd = rcp_cubic(det);
            const double U = -(u * rd), V = -(v * rd);
            // (the scheduling barriers keep the compiler from hoisting all broadcasts of a pair to its top: ~50 live scalar pairs, which
            // the big kernels this is inlined into spill through v_writelane and on into scratch)
#pragma unroll
            for (int k = c + 2; k < CH_NB; ++k) {
                X[k] = fma(lane_bcast(x1, k), V, fma(lane_bcast(x0, k), U, X[k]));
                if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        X[c + 1] = pw ? v : x1;
        S = (l16 == c) ? (pv ? x0 : 1.0) : S;
        S = (l16 == c + 1) ? (pw ? a * v : 1.0) : S;
    }
    // every pivot a > 0 and det > 0 (lane c: a; lane c + 1: a det) -- NaN and infinity fail
    const unsigned long long bad = __ballot(!(S > 0.0) || !isfinite(S));
    const double rs = rsqrt_cubic(S);
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) X[c] *= lane_bcast(rs, c);
    return (bad & 0xffffull) == 0;
}

typedef double chol_d4 __attribute__((ext_vector_type(4)));

// Development aid (tools/chol_test.hip -DXR_TL_TIMELINE): shader-clock phase sums of tl_chol as wavefront 0 sees them.
#ifndef XR_TL_CLK
#define XR_TL_CLK(slot, on)
#define XR_TL_CLK_RESET()
#endif

// Wavefront 0 only.  Factors the nb x nb diagonal block at (j0, j0) of the packed matrix A in place (A gets L)
// and leaves L^-1 (lower triangular, zero above the diagonal, identity-padded to 16x16) in Dinv.
// If `rhs` is given (the LAST panel of a system with one right-hand-side row), the inverse is not needed: the
// row segment rhs[j0 .. j0+nb) is forward-substituted in registers instead and Dinv is left untouched.
// Returns false (uniformly over the wavefront) on a non-positive pivot.
#ifdef XRHIP_KPROF
#define DGPROF(slot)                                   \
    do {                                               \
        if (dprof && lane == 0) {                      \
            const long long t_now = wall_clock64();    \
            dprof[slot] += t_now - t_dg;               \
            t_dg = t_now;                              \
        }                                              \
    } while (0)
#else
#define DGPROF(slot) \
    do {             \
    } while (0)
#endif
template <bool WITH_RHS>
__device__ __forceinline__ bool chol_diag_wave_t(double *A, int j0, int nb, double (*Dinv)[CH_NB + 1], int lane, double *rhs,
                                                 long long *dprof = nullptr) {
#ifdef XRHIP_KPROF
    long long t_dg = wall_clock64();
#endif
    // lanes 0-15: row `lane` of the block (rows >= nb are identity rows); riding rows: the right-hand-side segment in lane 16
    // (WITH_RHS) or the unit rows in lanes 16-31, which come out as the columns of L^-1 (X[r] = Linv[r][lane - 16])
    const int l16 = lane & 15;
    const bool is_row = lane < CH_NB, is_ride = WITH_RHS ? lane == CH_NB : (lane >= CH_NB && lane < 2 * CH_NB);
    double X[CH_NB];
#pragma unroll
    for (int k = 0; k < CH_NB; ++k) {
        double v = 0.0;
        if (is_row) v = (lane < nb && k <= lane) ? A[tri_idx(j0 + lane, j0 + k)] : ((k == lane) ? 1.0 : 0.0);
        else if (WITH_RHS) v = (is_ride && k < nb) ? rhs[j0 + k] : 0.0;
        else v = (is_ride && k == l16) ? 1.0 : 0.0;
        X[k] = v;
    }
    DGPROF(4);   // block load
    const bool ok = diag16_pivot_pairs(X, nb, lane);
    DGPROF(5);   // factorisation (+ inverse / substitution, riding)
    if (is_row) {
        if (lane < nb) {
#pragma unroll
            for (int k = 0; k < CH_NB; ++k)
                if (k <= lane) A[tri_idx(j0 + lane, j0 + k)] = X[k];
        }
    } else if (is_ride) {
#pragma unroll
        for (int k = 0; k < CH_NB; ++k) {
            if (WITH_RHS) {
                if (k < nb) rhs[j0 + k] = X[k];
            } else {
                Dinv[k][l16] = X[k];
            }
        }
    }
    DGPROF(6);   // write-back
    return ok;
}
__device__ __forceinline__ bool chol_diag_wave(double *A, int j0, int nb, double (*Dinv)[CH_NB + 1], int lane,
                                               double *rhs = nullptr, long long *dprof = nullptr) {
    return rhs ? chol_diag_wave_t<true>(A, j0, nb, Dinv, lane, rhs, dprof) : chol_diag_wave_t<false>(A, j0, nb, Dinv, lane, nullptr, dprof);
}

// In-place blocked Cholesky of the packed lower triangle A (n x n): on success A holds L.  Rows n .. nrows-1
// (nrows >= n, packed right behind the matrix) are carried along as right-hand sides: they end up holding
// L^-1 rhs.  Dinv is a [CH_NB][CH_NB+1] LDS scratch block, s_fail an LDS flag.  All threads of the workgroup
// must call.  Returns false (uniformly) if a non-positive pivot is met.
//
// One panel of lookahead: the 16x16 diagonal factorisation is a serial chain on ONE wavefront (~6 us) and used to
// leave the others idle, followed by a trailing update (~4 us at 150 unknowns) that left nothing for the next
// diagonal block to overlap with.  Now wavefront 0 updates the next diagonal tile (and, before the last panel, the
// right-hand-side tile that block needs) first and factors it while the other wavefronts finish the trailing update.
// Every tile is computed exactly as before, only by a different wavefront: results are bitwise unchanged.
__device__ __forceinline__ void chol_trailing_tile(double *A, int n, int nrows, int j0, int nb, int jb, int ti, int tj,
                                                   int r16, int q) {
    // A22 tile (ti, tj) -= P_ti P_tj^T, one tile = four v_mfma_f64_16x16x4_f64 (k = 16)
    const int gi = jb + 16 * ti + r16, gk = jb + 16 * tj + r16;
    const double *pa = A + tri_idx(min(gi, nrows - 1), j0) + q;
    const double *pb = A + tri_idx(min(gk, n - 1), j0) + q;
    const bool va = gi < nrows, vb = gk < n;
    chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const bool vk = 4 * s4 + q < nb;
        const double av = (va && vk) ? pa[4 * s4] : 0.0, bv = (vb && vk) ? pb[4 * s4] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oi = jb + 16 * ti + q + 4 * r, ok = jb + 16 * tj + r16;
        if (oi < nrows && ok < n && ok <= oi) A[tri_idx(oi, ok)] -= acc[r];
    }
}

struct CholNoSide {
    __device__ __forceinline__ void operator()() const {}
};
// `side`: work wavefront 1 does while wavefront 0 factors the FIRST diagonal block (a serial chain during which every other
// wavefront would only wait at the barrier) -- anything that does not touch A, Dinv or s_fail.
template <class Side = CholNoSide>
__device__ __forceinline__ bool chol_blocked(double *A, int n, int nrows, double (*Dinv)[CH_NB + 1], int *s_fail,
                                             long long *prof = nullptr, Side side = Side()) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int r16 = lane & 15, q = lane >> 4;
#ifdef XRHIP_KPROF
    long long t_prev = wall_clock64();
#endif
    if (tid == 0) *s_fail = 0;
    __syncthreads();
    double *rhs = A + tri_idx(n, 0);
    // ---- (1) diagonal block: factor + invert, wavefront 0, registers only.  The last panel of a system with a
    // single rhs row needs no inverse: that row is substituted in the same registers.
    if (n > 0 && wave == 0) {
        const int nb0 = min(CH_NB, n);
        const bool lwr0 = (nb0 >= n) && (nrows == n + 1);
        if (!chol_diag_wave(A, 0, nb0, Dinv, lane, lwr0 ? rhs : nullptr) && lane == 0) *s_fail = 1;
    } else if (wave == 1) {
        side();
    }
    __syncthreads();
    CHPROF(0);
    if (*s_fail) return false;
    for (int j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = min(CH_NB, n - j0);
        const int jb = j0 + nb;
        const bool last_with_rhs = (jb >= n) && (nrows == n + 1);
        if (jb >= nrows || last_with_rhs) break;
        // ---- (2) panel: X = P Linv^T for the rows below the block, 16-row tiles on the matrix cores.
        // A[i][k] from lane (i = lane & 15, k = lane >> 4), B[k][j] = Linv[j][k] from lane (k = lane >> 4, j = lane & 15).
        const int T = (nrows - jb + 15) >> 4;
        for (int t = wave; t < T; t += nw) {
            const int gi = jb + 16 * t + r16;
            const bool va = gi < nrows;
            const double *pa = A + tri_idx(min(gi, nrows - 1), j0) + q;
            chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kcol = 4 * s4 + q;
                const double av = (va && kcol < nb) ? pa[4 * s4] : 0.0;
                const double bv = Dinv[r16][kcol];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
            // D[(lane >> 4) + 4 r][lane & 15]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oi = jb + 16 * t + q + 4 * r;
                if (oi < nrows && r16 < nb) A[tri_idx(oi, j0 + r16)] = acc[r];
            }
        }
        __syncthreads();
        CHPROF(1);
        // ---- (3) trailing update A22 -= P P^T over the 16x16 tiles of the lower triangle (and of the rhs rows),
        // overlapped with (1) of the next panel
        const bool has_next = jb < n;
        const int nb1 = has_next ? min(CH_NB, n - jb) : 0;
        const bool lwr1 = has_next && (jb + nb1 >= n) && (nrows == n + 1);
        const int rhs_ti = (n - jb) >> 4;            // tile row that holds row n (the first rhs row)
        const bool look = has_next && nw > 1;        // wavefront 0 leaves the bulk of the tiles to the others
        if (look && wave == 0) {
            chol_trailing_tile(A, n, nrows, j0, nb, jb, 0, 0, r16, q);
            if (lwr1 && rhs_ti != 0) chol_trailing_tile(A, n, nrows, j0, nb, jb, rhs_ti, 0, r16, q);
            __threadfence_block();   // other lanes of this wavefront read those entries next (A may be in global memory)
            if (!chol_diag_wave(A, jb, nb1, Dinv, lane, lwr1 ? rhs : nullptr, prof) && lane == 0) *s_fail = 1;
        } else {
            // dealt round-robin with a counter that wraps (`t % workers` is a software division per tile visited, and every
            // wavefront visits all of them)
            const int workers = look ? nw - 1 : nw, me = look ? wave - 1 : wave;
            int turn = 0;
            for (int ti = 0; ti < T; ++ti)
                for (int tj = 0; tj <= ti; ++tj) {
                    if (jb + 16 * tj >= n) continue;   // rhs rows have no columns of their own
                    if (look && tj == 0 && (ti == 0 || (lwr1 && ti == rhs_ti))) continue;   // wavefront 0's tiles
                    if (turn == me) chol_trailing_tile(A, n, nrows, j0, nb, jb, ti, tj, r16, q);
                    turn = (turn + 1 == workers) ? 0 : turn + 1;
                }
        }
        __syncthreads();
        if (has_next && !look) {   // single-wavefront workgroups: no overlap to be had
            if (!chol_diag_wave(A, jb, nb1, Dinv, lane, lwr1 ? rhs : nullptr) && lane == 0) *s_fail = 1;
            __syncthreads();
        }
        CHPROF(2);
        if (*s_fail) return false;
    }
    return true;
}

// y <- L^-1 y  (forward) with L packed lower in A.  All threads must call.  The 16x16 diagonal systems are
// solved by wavefront 0 in registers (lane k owns row k of the block and y[j0 + k]).
__device__ __forceinline__ void trsv_lower(const double *A, int n, double *y) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    for (int j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = min(CH_NB, n - j0);
        if (wave == 0) {
            double row[CH_NB], dl = 1.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                row[c] = (lane < nb && c <= lane) ? A[tri_idx(j0 + lane, j0 + c)] : ((c == lane) ? 1.0 : 0.0);
                if (c == lane) dl = row[c];
            }
            const double dinv = 1.0 / dl;
            double r = (lane < nb) ? y[j0 + lane] : 0.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                const double xc = lane_bcast(r, c) * lane_bcast(dinv, c);
                if (lane == c) r = xc;
                else if (lane > c) r -= row[c] * xc;
            }
            if (lane < nb) y[j0 + lane] = r;
        }
        __syncthreads();
        const int jb = j0 + nb;
        for (int i = jb + tid; i < n; i += nt) {
            const double *rowp = A + tri_idx(i, j0);
            double s = 0;
            for (int c = 0; c < nb; ++c) s += rowp[c] * y[j0 + c];
            y[i] -= s;
        }
        __syncthreads();
    }
}

// y <- L^-T y  (backward).  All threads must call.  Lane k of wavefront 0 owns column k of the diagonal block.
__device__ __forceinline__ void trsv_lower_t(const double *A, int n, double *y) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int last = ((n - 1) / CH_NB) * CH_NB;
    for (int j0 = last; j0 >= 0; j0 -= CH_NB) {
        const int nb = min(CH_NB, n - j0);
        if (wave == 0) {
            double col[CH_NB], dl = 1.0;   // col[c] = L[j0 + c][j0 + lane], c >= lane
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                col[c] = (c < nb && lane <= c) ? A[tri_idx(j0 + c, j0 + lane)] : ((c == lane) ? 1.0 : 0.0);
                if (c == lane) dl = col[c];
            }
            const double dinv = 1.0 / dl;
            double r = (lane < nb) ? y[j0 + lane] : 0.0;
#pragma unroll
            for (int c = CH_NB - 1; c >= 0; --c) {
                const double xc = lane_bcast(r, c) * lane_bcast(dinv, c);
                if (lane == c) r = xc;
                else if (lane < c) r -= col[c] * xc;
            }
            if (lane < nb) y[j0 + lane] = r;
        }
        __syncthreads();
        for (int i = tid; i < j0; i += nt) {
            double s = 0;
            for (int c = 0; c < nb; ++c) s += A[tri_idx(j0 + c, i)] * y[j0 + c];
            y[i] -= s;
        }
        __syncthreads();
    }
}


// X <- L^-1 for a packed lower-triangular L (n x n, tri_idx) in LDS; X is a second packed triangle.  All threads must call.
// Used by the marginalisation's eigenvalue guard (km_chol): |L^-1|_F^2 = trace(A^-1).  Round 4 did one forward substitution per THREAD
// (column j of the inverse: a dependent chain of up to n^2 / 2 multiply-adds on one lane -- most of km_chol's 126 us); here 16-wide
// blocks: the diagonal blocks by one wavefront each (lane = column, the block's entries are wave-uniform LDS reads), then block row
// after block row X_ij = -X_ii (sum_{k=j}^{i-1} L_ik X_kj) on the f64 matrix cores, one 16x16 tile per wavefront at a time.
__device__ __forceinline__ void tri_inverse_blocked(const double *L, double *X, int n) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int T = (n + 15) >> 4;
    for (int b = wave; b < T; b += nw) {
        const int j0 = 16 * b, nb = min(16, n - j0);
        double y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (r == r16) ? 1.0 : 0.0;
            const bool row = r < nb;
            const double *Lr = L + tri_idx(j0 + (row ? r : 0), j0);
#pragma unroll
            for (int k = 0; k < r; ++k) acc = fma(-(row ? Lr[k] : 0.0), y[k], acc);
            y[r] = row ? acc / Lr[r] : acc;
        }
        if (lane < nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r >= lane && r < nb) X[tri_idx(j0 + r, j0 + lane)] = y[r];
        }
    }
    __syncthreads();
    for (int i = 1; i < T; ++i) {
        for (int j = wave; j < i; j += nw) {
            chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
            const int gi = 16 * i + r16;
            const bool vi = gi < n;
            for (int k = j; k < i; ++k) {
                const double *pa = L + tri_idx(vi ? gi : 0, 16 * k + q);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int kk = 4 * s4 + q;   // row of X_kj, column of L_ik
                    const double av = vi ? pa[4 * s4] : 0.0;
                    const bool vb = (k > j) || (r16 <= kk);   // X_jj is lower triangular
                    const double bv = vb ? X[tri_idx(16 * k + kk, 16 * j + (vb ? r16 : 0))] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
            // S[(q + 4 r)][r16] sits in acc[r]: exactly the B operand (k = 4 s4 + q, n = r16) of the second product's step s4 = r
            chol_d4 out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kk = 4 * s4 + q;
                const bool va = vi && kk <= r16;
                const double av = va ? X[tri_idx(gi, 16 * i + kk)] : 0.0;   // X_ii[r16][kk]
                out = __builtin_amdgcn_mfma_f64_16x16x4f64(av, acc[s4], out, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oi = 16 * i + q + 4 * r;
                if (oi < n) X[tri_idx(oi, 16 * j + r16)] = -out[r];
            }
        }
        __syncthreads();
    }
}

// =====================================================================================================================
// Tiled variant (round 3; round 6: block inverses kept IN the diagonal tiles) -- for systems that live in LDS.
//
// Why a second layout.  In the packed triangle the 16x16 blocks the matrix cores consume have a different row stride in every
// row (bank conflicts on every operand load), the diagonal block's load and write-back are sixteen predicated accesses each, and
// the back-substitution has to re-solve every diagonal block as a 16-step chain because the inverse of a block is gone once the
// next panel has used it.
//
// Layout: the lower triangle in 16x16 tiles, tile (ti, tj), tj <= ti, at (ti (ti + 1) / 2 + tj) * TL_TILE doubles; inside a tile
// element (r, c) at c * TL_LD + r with TL_LD = 17: "row = lane" accesses (the diagonal block's rows, the matrix-core operands)
// and "column = lane" accesses (the transposed reads of the back-substitution) are both free of bank conflicts.  Rows >= nrows are
// zero, the padding diagonal is 1: nothing in the loops is predicated.  A right-hand side rides along as row n (nrows = n + 1): the
// factorisation leaves L^-1 rhs there.
//
// What the factorisation leaves behind: the tiles BELOW the diagonal hold L; a DIAGONAL tile holds the transpose of its block's
// inverse, element (r, c) = Linv[c][r] (upper triangular, exact zeros below the diagonal) -- nobody reads the diagonal blocks of L
// again (the right-hand side was substituted on the way), while the inverse is what the panel product P Linv^T and the
// back-substitution (one 16x16 matrix-vector product per block instead of a 16-step chain) consume.  Rows of the last diagonal
// tile that are not pivots (the right-hand-side row when n is not a multiple of 16, padding) keep their own entries.
//
// Diagonal block (tl_diag_wave): lanes 0-15 hold the rows of the block, lanes 16-31 unit rows that come out as the columns of the
// inverse (diag16_pivot_pairs); every lane loads X[k] from base + k TL_LD and stores it there again, only the base differs per lane
// group (the tile / an identity tile), so load and write-back are sixteen unpredicated LDS instructions each.
constexpr int TL_LD = 17, TL_TILE = 16 * TL_LD;
__host__ __device__ __forceinline__ int tl_tile_rows(int nrows) { return (nrows + 15) >> 4; }
__host__ __device__ __forceinline__ int tl_doubles(int nrows) {
    const int T = tl_tile_rows(nrows);
    return T * (T + 1) / 2 * TL_TILE;
}
__host__ __device__ __forceinline__ int tl_tile(int ti, int tj) { return (ti * (ti + 1) / 2 + tj) * TL_TILE; }
__host__ __device__ __forceinline__ int tl_idx(int i, int j) { return tl_tile(i >> 4, j >> 4) + (j & 15) * TL_LD + (i & 15); }   // j <= i

// zero fill with an identity padding diagonal (rows / columns >= n of the diagonal tiles); all threads
__device__ __forceinline__ void tl_clear(double *A, int n, int nrows) {
    const int T = tl_tile_rows(nrows), total = T * (T + 1) / 2 * TL_TILE;
    for (int e = threadIdx.x; e < total; e += blockDim.x) A[e] = 0.0;
    __syncthreads();
    for (int i = n + threadIdx.x; i < 16 * T; i += blockDim.x) A[tl_idx(i, i)] = 1.0;
}

// One wavefront.  Factors the diagonal tile `tile` and replaces it by the transposed inverse of its factor (see above).  ident: an
// identity tile (element (r, c) at c * TL_LD + r) in LDS.  nb: rows of the block that are pivots; rows nb.. (a right-hand-side row,
// padding) take part in the column operations, are never pivots and are written back as rows.
__device__ __forceinline__ bool tl_diag_wave(double *tile, int nb, const double *ident, int lane, long long *dprof = nullptr) {
#ifdef XRHIP_KPROF
    long long t_dg = wall_clock64();
#endif
    const int l16 = lane & 15;
    const double *src = (lane < CH_NB ? tile : ident) + l16;
    double X[CH_NB];
#pragma unroll
    for (int k = 0; k < CH_NB; ++k) X[k] = src[k * TL_LD];
    DGPROF(4);   // block load
    XR_TL_CLK(5, true);
    const bool ok = diag16_pivot_pairs(X, nb, lane);
    DGPROF(5);   // factorisation + inverse
    XR_TL_CLK(6, true);
    // row r of the tile: for a pivot row the unit row that rode in lane 16 + r (now Linv[.][r], i.e. row r of the transposed inverse),
    // otherwise the row itself (lane r)
    if (lane < CH_NB ? lane >= nb : lane < CH_NB + nb) {
        double *dst = tile + l16;
#pragma unroll
        for (int k = 0; k < CH_NB; ++k) dst[k * TL_LD] = X[k];
    }
    DGPROF(6);   // write-back
    XR_TL_CLK(7, true);
    return ok;
}

// Matrix-core operand of a panel tile: lane (r16 = lane & 15, q = lane >> 4) holds P[r16][4 s + q], s = 0 .. 3 -- as the A operand
// (rows r16) and as the B operand (columns r16) alike.  It is also EXACTLY what tl_panel_tile's accumulator holds, so a panel tile goes
// from the product that made it into the products that consume it without touching LDS.
__device__ __forceinline__ chol_d4 tl_load_operand(const double *tile, int r16, int q) {
    chol_d4 v;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) v[s4] = tile[(4 * s4 + q) * TL_LD + r16];
    return v;
}
// W = P Linv^T for the panel tile `pt` below the diagonal tile `dt` (which holds Linv^T: Linv[c][k] at (k, c)), computed as the
// transposed product W^T = Linv P^T: the accumulator's lane layout D[(lane >> 4) + 4 r][lane & 15] = W[r16][q + 4 r] is a
// column-major store and the operand layout of the products that follow.  The tile is overwritten by W; W is returned as well.
__device__ __forceinline__ chol_d4 tl_panel_tile(const double *dt, double *pt, int r16, int q) {
    const chol_d4 pv = tl_load_operand(pt, r16, q);
    chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dt[r16 * TL_LD + 4 * s4 + q], pv[s4], acc, 0, 0, 0);   // D'[c][i] = sum_k Linv[c][k] P[i][k]
#pragma unroll
    for (int r = 0; r < 4; ++r) pt[(q + 4 * r) * TL_LD + r16] = acc[r];   // element (row r16, column q + 4 r); every lane read its operand above
    return acc;
}
// acc += W_tj W_ti^T as the transposed product D'[j][i] = sum_k W_tj[j][k] W_ti[i][k] (operands: see tl_load_operand)
__device__ __forceinline__ chol_d4 tl_mac_tile(chol_d4 acc, const chol_d4 wj, const chol_d4 wi) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wj[s4], wi[s4], acc, 0, 0, 0);
    return acc;
}
// tile -= acc (accumulator layout: element (row r16, column q + 4 r) in acc[r])
__device__ __forceinline__ void tl_sub_tile(double *tile, const chol_d4 acc, int r16, int q) {
    double o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = tile[(q + 4 * r) * TL_LD + r16];
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(q + 4 * r) * TL_LD + r16] = o[r] - acc[r];
}

// In-place blocked Cholesky in the tiled layout; rows n .. nrows-1 ride along as right-hand sides.  ident: TL_TILE doubles of LDS
// scratch (becomes an identity tile).  All threads must call; returns false (uniformly) on a non-positive pivot.  `side`: see
// chol_blocked.
//
// Schedule (round 6).  Wavefront 0 walks the critical path and nothing else: block j -> [barrier] -> the panel tile right below it and,
// straight from that product's registers, the update of the NEXT diagonal tile -> [barrier] -> block j + 1.  The other wavefronts, per
// step j: their share of the panel; then, while wavefront 0 factors block j + 1, (a) the update of tile column j + 1 by step j (what
// the next panel needs) and (b) the update of tile column j + 2 by ALL steps 0 .. j at once, accumulated in the matrix cores'
// registers (left-looking).  Right-looking -- every tile behind the panel updated at every step -- made the first steps five times as
// long as the last ones (55 tile products at step 0 of an 11-tile system, 1 at step 9) with a workgroup barrier per step, so wavefront 0
// waited for the trailing update early on and the workers idled later; left-looking is (T - j - 2)(j + 2) products per step, 18 .. 30 ..
// 10, and hides behind the diagonal blocks throughout.  f64 matrix-core products cost 80 cycles per 16x16x4 whether dependent or not
// (tools/issue.hip): a tile product is 320 cycles of its SIMD's matrix pipe, ~220 of them per system of 165 unknowns.
template <class Side = CholNoSide>
__device__ __forceinline__ bool tl_chol(double *A, int n, int nrows, double *ident, int *s_fail, long long *prof = nullptr,
                                        Side side = Side()) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int T = tl_tile_rows(nrows), Tm = tl_tile_rows(n);
#ifdef XRHIP_KPROF
    long long t_prev = wall_clock64();
#endif
    if (tid == 0) *s_fail = 0;
    for (int e = tid; e < TL_TILE; e += nt) ident[e] = (e / TL_LD == e % TL_LD) ? 1.0 : 0.0;
    __syncthreads();
    XR_TL_CLK_RESET();
    if (n > 0 && wave == 0) {
        if (!tl_diag_wave(A + tl_tile(0, 0), min(CH_NB, n), ident, lane) && lane == 0) *s_fail = 1;
    } else if (wave == 1) {
        side();
    }
    XR_TL_CLK(0, wave == 0);
    __syncthreads();
    XR_TL_CLK(1, wave == 0);
    CHPROF(0);
    if (*s_fail) return false;
    const int workers = nw > 1 ? nw - 1 : 1, me = nw > 1 ? wave - 1 : 0;   // wavefront 0 of a multi-wavefront workgroup is no worker
    const bool is_worker = nw == 1 || wave > 0;
    for (int j = 0; j < Tm; ++j) {
        if (j + 1 >= T) break;   // nothing below this block
        const double *dt = A + tl_tile(j, j);
        const bool has_next = j + 1 < Tm;   // tile row j + 1 is a matrix row (a diagonal block follows), not only right-hand sides
        // ---- panel.  Row j + 1 is wavefront 0's, together with the next diagonal tile's update by this step.
        if (wave == 0) {
            const chol_d4 w = tl_panel_tile(dt, A + tl_tile(j + 1, j), r16, q);
            if (has_next) tl_sub_tile(A + tl_tile(j + 1, j + 1), tl_mac_tile(chol_d4{0.0, 0.0, 0.0, 0.0}, w, w), r16, q);
        }
        if (is_worker)
            for (int t = j + 2 + me; t < T; t += workers) tl_panel_tile(dt, A + tl_tile(t, j), r16, q);
        XR_TL_CLK(2, wave == 0);
        __syncthreads();
        XR_TL_CLK(3, wave == 0);
        CHPROF(1);
        // ---- wavefront 0: the next block.  Workers: (a) column j + 1 by step j, (b) column j + 2 by steps 0 .. j.
        if (wave == 0 && has_next) {
            XR_TL_CLK(4, true);
            if (!tl_diag_wave(A + tl_tile(j + 1, j + 1), min(CH_NB, n - 16 * (j + 1)), ident, lane, prof) && lane == 0) *s_fail = 1;
        }
        if (is_worker && has_next) {
            int turn = 0;   // tiles dealt round-robin, (b) first (the heavier ones), with a counter that wraps
            if (j + 2 < Tm) {
                for (int t = j + 2; t < T; ++t) {
                    if (turn == me) {
                        chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
                        for (int s_ = 0; s_ <= j; ++s_)
                            acc = tl_mac_tile(acc, tl_load_operand(A + tl_tile(j + 2, s_), r16, q), tl_load_operand(A + tl_tile(t, s_), r16, q));
                        tl_sub_tile(A + tl_tile(t, j + 2), acc, r16, q);
                    }
                    turn = (turn + 1 == workers) ? 0 : turn + 1;
                }
            }
            const chol_d4 wd = tl_load_operand(A + tl_tile(j + 1, j), r16, q);
            for (int t = j + 2; t < T; ++t) {
                if (turn == me)
                    tl_sub_tile(A + tl_tile(t, j + 1), tl_mac_tile(chol_d4{0.0, 0.0, 0.0, 0.0}, wd, tl_load_operand(A + tl_tile(t, j), r16, q)), r16, q);
                turn = (turn + 1 == workers) ? 0 : turn + 1;
            }
        }
        XR_TL_CLK(9, wave == 0);
        __syncthreads();
        XR_TL_CLK(8, wave == 0);
        CHPROF(2);
        if (*s_fail) return false;
    }
    return true;
}

// y <- L^-T y with what tl_chol left behind (L below the diagonal, transposed block inverses on it).  y: a vector of
// 16 * tile_rows(n) doubles in LDS whose entries >= n are ZERO.  All threads must call.
//
// Per block j, last to first: x_j = Linv_j^T y_j, then y_i -= L_ji^T x_j for the blocks i above.  One workgroup barrier per block
// (round 6; three before): wavefront 0 forms x_j (lane = (row r, quarter q): four products, the quarters added by shuffles), hands it to
// the others through y and the barrier, and takes block j - 1's share of the update -- the next block it needs -- itself, from its
// registers, while the other wavefronts update the blocks above that.
__device__ __forceinline__ void tl_trsv_t(const double *A, int n, double *y) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int Tm = tl_tile_rows(n);
    const int r = lane & 15, q = lane >> 4;
    // wavefront 0: what block j + 1's solution takes off y_j -- lane (r, any quarter): sum_c L[16 (j+1) + c][16 j + r] x_{j+1}[c].  Kept in
    // registers, not subtracted in LDS: the other wavefronts are still updating y_j by x_{j+2} when it is formed.
    double pend = 0.0;
    for (int j = Tm - 1; j >= 0; --j) {
        if (wave == 0) {
            // x_r = sum_c Linv[c][r] (y_c - pend_c): row r of the diagonal tile (zeros left of the diagonal; whatever a non-pivot column
            // holds meets a zero of y); lane (r, q) takes c = 4 q .. 4 q + 3
            const double *tile = A + tl_tile(j, j) + r + 4 * q * TL_LD;
            const double *yj = y + 16 * j + 4 * q;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += tile[k * TL_LD] * (yj[k] - __shfl(pend, 4 * q + k));
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const double xr = (16 * j + r < n) ? s : 0.0;
            if (q == 0) y[16 * j + r] = xr;
            pend = 0.0;
            if (j > 0) {
                const double *col = A + tl_tile(j, j - 1) + r * TL_LD + 4 * q;   // element (c, r) of tile (j, j - 1): L[16 j + c][16 (j - 1) + r]
                double u = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) u += col[k] * __shfl(xr, 4 * q + k);
                u += __shfl_xor(u, 16);
                u += __shfl_xor(u, 32);
                pend = u;
            }
        }
        __syncthreads();
        // y_i -= sum_c L[16 j + c][i] x_c for the blocks above j - 1, by the other wavefronts (by this one when it is alone): thread i
        // reads element (c, i & 15) of tile (j, i >> 4), c = 0 .. 15
        const int first = nt > 64 ? 64 : 0;
        for (int i = tid - first; i >= 0 && i < 16 * (j - 1); i += nt - first) {
            const double *col = A + tl_tile(j, i >> 4) + (i & 15) * TL_LD;
            const double *xj = y + 16 * j;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) s += col[c] * xj[c];
            y[i] -= s;
        }
        // (no second barrier: block j - 1 is solved from y_{j-1} as the barrier above left it -- the others' updates by x_{j+1}, ... were
        // complete there -- minus `pend`; what the others write now are rows above block j - 1, read after the next barrier)
    }
    __syncthreads();
}

}   // namespace xrhip
