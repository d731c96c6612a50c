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

typedef double chol_d4 __attribute__((ext_vector_type(4)));

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
    double x[CH_NB];   // row `lane` of the block; rows >= nb are identity rows
#pragma unroll
    for (int k = 0; k < CH_NB; ++k) x[k] = (lane < nb && k <= lane) ? A[tri_idx(j0 + lane, j0 + k)] : ((k == lane) ? 1.0 : 0.0);
    DGPROF(4);   // block load
    double dinv[CH_NB];
    double xi[CH_NB];   // column `lane` of L^-1: xi[r] = Linv[r][lane]
    // acc[r] = delta(r, lane) - sum_{k < r} L[r][k] Linv[k][lane], built up column by column (see below)
    double acc[CH_NB];
#pragma unroll
    for (int r = 0; r < CH_NB; ++r) acc[r] = (r == lane) ? 1.0 : 0.0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        double dcc = lane_bcast(x[c], c);
        if (!(dcc > 0.0) || !isfinite(dcc)) {
            ok = false;
            dcc = 1.0;
        }
        // sqrt(d) and 1/sqrt(d) together from the hardware reciprocal square root (v_rsq_f64, ~27 bits) and two coupled
        // Newton steps -- the library sqrt followed by a division is ~3x as long, and this sits on the serial chain of
        // every panel.  Pivots here are O(1) after Jacobi scaling (1e30 at most, in the marginalisation): no denormals.
        double dd, hh;
        {
            const double r0 = __builtin_amdgcn_rsq(dcc);
            dd = dcc * r0;
            hh = 0.5 * r0;
            double e = fma(-hh, dd, 0.5);
            dd = fma(dd, e, dd);
            hh = fma(hh, e, hh);
            e = fma(-hh, dd, 0.5);
            dd = fma(dd, e, dd);
            hh = fma(hh, e, hh);
            const double res = fma(-dd, dd, dcc);
            dd = fma(res, hh, dd);
        }
        dinv[c] = hh + hh;
        x[c] = (lane == c) ? dd : x[c] * dinv[c];
        // Row c of L^-1 is complete once the columns before c have been through (forward substitution, terms added in the
        // order k = 0, 1, ...).  The inverse is built RIGHT-looking, beside the factorisation: the broadcast L[k][c] that
        // updates column k of the block also carries row c of the inverse into acc[k].  (Left-looking -- row c summed up
        // when column c is reached -- reads the same broadcasts a second time, up to fifteen columns later: the compiler
        // keeps all 120 of them alive in scalar registers and spills them through v_writelane.)  Same products, same order of
        // additions: the same bits.
        if (!WITH_RHS) xi[c] = (c < lane) ? 0.0 : acc[c] * dinv[c];
#pragma unroll
        for (int k = c + 1; k < CH_NB; ++k) {
            const double lkc = lane_bcast(x[c], k);   // L[k][c]
            x[k] -= x[c] * lkc;                        // meaningful for lane >= k (lower triangle)
            if (!WITH_RHS) acc[k] -= lkc * xi[c];      // L[k][c] * Linv[c][lane]
        }
    }
    DGPROF(5);   // factorisation (+ inverse)
    if (WITH_RHS) {
        double r = (lane < nb) ? rhs[j0 + lane] : 0.0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) {
            const double xc = lane_bcast(r, c) * dinv[c];
            if (lane == c) r = xc;
            else if (lane > c) r -= x[c] * xc;   // L[lane][c]
        }
        if (lane < nb) {
            rhs[j0 + lane] = r;
#pragma unroll
            for (int k = 0; k < CH_NB; ++k)
                if (k <= lane) A[tri_idx(j0 + lane, j0 + k)] = x[k];
        }
        return ok;
    }
    // Every lane stores its column of the inverse -- lanes 16..63 into the padding column of Dinv.  Under `if (lane < 16)` the
    // compiler sinks the whole inverse into that branch: it then runs AFTER the factorisation instead of beside it, from
    // broadcasts it has kept in VGPR lanes (v_writelane / v_readlane, a third of this routine's instruction stream).
    {
        const int col = lane < CH_NB ? lane : CH_NB;
#pragma unroll
        for (int r = 0; r < CH_NB; ++r) Dinv[r][col] = xi[r];
    }
    if (lane < CH_NB) {
#pragma unroll
        for (int k = 0; k < CH_NB; ++k)
            if (lane < nb && k <= lane) A[tri_idx(j0 + lane, j0 + k)] = x[k];
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
            const int workers = look ? nw - 1 : nw, me = look ? wave - 1 : wave;
            int t = 0;
            for (int ti = 0; ti < T; ++ti)
                for (int tj = 0; tj <= ti; ++tj) {
                    if (jb + 16 * tj >= n) continue;   // rhs rows have no columns of their own
                    if (look && tj == 0 && (ti == 0 || (lwr1 && ti == rhs_ti))) continue;   // wavefront 0's tiles
                    if (t++ % workers != me) continue;
                    chol_trailing_tile(A, n, nrows, j0, nb, jb, ti, tj, r16, q);
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
// Tiled variant (round 3) -- for systems that live in LDS.
//
// Why a second layout.  In the packed triangle the 16x16 blocks the matrix cores consume have a different row stride in every
// row (bank conflicts on every operand load), the diagonal block's load and write-back are sixteen predicated accesses each, and
// the back-substitution has to re-solve every diagonal block as a 16-step dependent chain because the inverse of a block is
// gone once the next panel has used it.  In-kernel timers of round 2 (profiles/r02_kprof_v39.md): of a panel step's 5.8 us the
// serial diagonal block was 3.9 (0.5 load + 2.0 factor/invert + 1.4 write-back), the substitution another 1.6 us per block.
//
// Layout: the lower triangle in 16x16 tiles, tile (ti, tj), tj <= ti, at (ti (ti + 1) / 2 + tj) * TL_TILE doubles; inside a tile
// element (r, c) at c * TL_LD + r with TL_LD = 17: "row = lane" reads (the matrix-core operands, the diagonal block) and "column =
// lane" reads (the transposed accesses of the back-substitution) are both free of bank conflicts.  Rows >= nrows are zero, the
// padding diagonal is 1: nothing in the loops is predicated.  A right-hand side rides along as row n (nrows = n + 1): the
// factorisation leaves L^-1 rhs there.  The strictly upper triangle of a DIAGONAL tile -- unused by L -- keeps the transposed
// strictly lower triangle of that block's inverse, its diagonal goes to dinv[tile][16]: the back-substitution is then one
// 16x16 matrix-vector product per block instead of a 16-step chain.
//
// Diagonal block (tl_diag_wave): lanes 0-15 hold the rows of the block, lanes 16-31 the columns of the inverse under
// construction, in the SAME registers: the rank-1 update of column c is one FMA per later column for both (the inverse was a second
// FMA on the same lanes before).  Same products and sums as chol_diag_wave_t, in the same order: the same bits.
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

// Wavefront 0.  Factors the diagonal tile `tile` (in place: L in the lower triangle), leaves the block's inverse in Dinv (for the
// panel product) and, transposed, in the tile's upper triangle + dinv16 (for the back-substitution).  nb: rows of the block that are
// matrix rows; rows nb.. (a right-hand-side row, padding) take part in the column operations but are never pivots.
__device__ __forceinline__ bool tl_diag_wave(double *tile, int nb, double (*Dinv)[CH_NB + 1], double *dinv16, int lane,
                                             long long *dprof = nullptr) {
#ifdef XRHIP_KPROF
    long long t_dg = wall_clock64();
#endif
    const int l16 = lane & 15;
    const bool is_row = lane < 16, is_inv = lane >= 16 && lane < 32;
    // lanes 0-15: X[k] = row l16 of the block (lower triangle); lanes 16-31: X[k] = delta(k, l16), the running forward substitution
    // of unit vector l16 (column l16 of the inverse once divided through)
    double X[CH_NB];
#pragma unroll
    for (int k = 0; k < CH_NB; ++k) {
        const double v = tile[k * TL_LD + l16];
        X[k] = is_row ? (k <= l16 ? v : 0.0) : ((k == l16) ? 1.0 : 0.0);
    }
    DGPROF(4);   // block load
    bool ok = true;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        double dcc = lane_bcast(X[c], c);
        if (c >= nb) dcc = 1.0;   // not a pivot: the row only rides along
        if (!(dcc > 0.0) || !isfinite(dcc)) {
            ok = false;
            dcc = 1.0;
        }
        double dd, hh;
        {   // sqrt(d) and 1/sqrt(d): v_rsq_f64 + two coupled Newton steps (see chol_diag_wave_t)
            const double r0 = __builtin_amdgcn_rsq(dcc);
            dd = dcc * r0;
            hh = 0.5 * r0;
            double e = fma(-hh, dd, 0.5);
            dd = fma(dd, e, dd);
            hh = fma(hh, e, hh);
            e = fma(-hh, dd, 0.5);
            dd = fma(dd, e, dd);
            hh = fma(hh, e, hh);
            const double res = fma(-dd, dd, dcc);
            dd = fma(res, hh, dd);
        }
        const double dinv = hh + hh;
        // row lanes: L[l][c] = x / d (the pivot row itself: d); inverse lanes: Linv[c][m] = acc / d
        X[c] = (lane == c) ? dd : X[c] * dinv;
#pragma unroll
        for (int k = c + 1; k < CH_NB; ++k) {
            const double lkc = lane_bcast(X[c], k);   // L[k][c] (held by row lane k)
            X[k] -= X[c] * lkc;                        // rows: A[l][k] -= L[l][c] L[k][c];  inverse: acc[k] -= Linv[c][m] L[k][c]
        }
    }
    DGPROF(5);   // factorisation + inverse
    // write-back.  Row lane l: L[l][k], k <= l, to (l, k).  Inverse lane m (column m of the inverse, X[r] = Linv[r][m], r >= m): Dinv[r][m]
    // for the panel product; transposed into the strictly upper triangle, (m, r) for r > m; the diagonal to dinv16.
    double *dummy = &Dinv[0][CH_NB];   // padding column: a harmless target for lanes that have nothing to store
#pragma unroll
    for (int k = 0; k < CH_NB; ++k) {
        double *dst = dummy;
        if (is_row && k <= l16) dst = tile + k * TL_LD + l16;
        else if (is_inv) dst = &Dinv[k][l16];
        *dst = X[k];
    }
#pragma unroll
    for (int r = 1; r < CH_NB; ++r) {
        double *dst = dummy;
        if (is_inv && r > l16) dst = tile + r * TL_LD + l16;
        *dst = X[r];
    }
    {
        // the diagonal of the inverse: lane 16 + m holds it in X[m] (register index = lane index): select without dynamic indexing
        double dgl = 0.0;
#pragma unroll
        for (int k = 0; k < CH_NB; ++k) dgl = (k == l16) ? X[k] : dgl;
        if (is_inv) dinv16[l16] = dgl;
    }
    DGPROF(6);   // write-back
    return ok;
}

// tile (ti, tj) -= P_ti P_tj^T with the panel tiles of tile column tc; computed as the TRANSPOSED product (operands swapped) so that
// the accumulator's lane layout, D[(lane >> 4) + 4 r][lane & 15], is a column-major store
__device__ __forceinline__ void tl_trailing_tile(double *A, int tc, int ti, int tj, int r16, int q) {
    const double *pi = A + tl_tile(ti, tc) + r16, *pj = A + tl_tile(tj, tc) + r16;
    double *out = A + tl_tile(ti, tj) + r16;
    chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int kc = (4 * s4 + q) * TL_LD;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[kc], pi[kc], acc, 0, 0, 0);   // D'[j][i] = sum_k P_tj[j][k] P_ti[i][k]
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(q + 4 * r) * TL_LD] -= acc[r];   // element (row r16, column q + 4 r)
}

// In-place blocked Cholesky in the tiled layout; rows n .. nrows-1 ride along as right-hand sides.  dinv: [tile rows][16].
// All threads must call; returns false (uniformly) on a non-positive pivot.  `side`: see chol_blocked.
template <class Side = CholNoSide>
__device__ __forceinline__ bool tl_chol(double *A, int n, int nrows, double (*Dinv)[CH_NB + 1], double *dinv, int *s_fail,
                                        long long *prof = nullptr, Side side = Side()) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int T = tl_tile_rows(nrows), Tm = tl_tile_rows(n);
#ifdef XRHIP_KPROF
    long long t_prev = wall_clock64();
#endif
    if (tid == 0) *s_fail = 0;
    __syncthreads();
    if (n > 0 && wave == 0) {
        if (!tl_diag_wave(A + tl_tile(0, 0), min(CH_NB, n), Dinv, dinv, lane) && lane == 0) *s_fail = 1;
    } else if (wave == 1) {
        side();
    }
    __syncthreads();
    CHPROF(0);
    if (*s_fail) return false;
    for (int j = 0; j < Tm; ++j) {
        if (j + 1 >= T) break;   // nothing below this block
        // ---- panel: X = P Linv^T for the tile rows below, as X^T = Linv P^T (column-major store of the accumulator)
        for (int t = j + 1 + wave; t < T; t += nw) {
            double *pt = A + tl_tile(t, j) + r16;
            chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kcol = 4 * s4 + q;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Dinv[r16][kcol], pt[kcol * TL_LD], acc, 0, 0, 0);   // D'[c][i] = sum_k Linv[c][k] P[i][k]
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // every lane has read the tile before anybody overwrites it
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[(q + 4 * r) * TL_LD] = acc[r];   // element (row r16, column q + 4 r)
        }
        __syncthreads();
        CHPROF(1);
        // ---- trailing update over the tiles (ti, tj), j < tj <= ti < T, tj < Tm; wavefront 0 takes the next diagonal tile first and
        // factors it while the others finish (one panel of look-ahead)
        const bool has_next = j + 1 < Tm;
        const bool look = has_next && nw > 1;
        if (look && wave == 0) {
            tl_trailing_tile(A, j, j + 1, j + 1, r16, q);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!tl_diag_wave(A + tl_tile(j + 1, j + 1), min(CH_NB, n - 16 * (j + 1)), Dinv, dinv + 16 * (j + 1), lane, prof) && lane == 0)
                *s_fail = 1;
        } else {
            const int workers = look ? nw - 1 : nw, me = look ? wave - 1 : wave;
            int t = 0;
            for (int ti = j + 1; ti < T; ++ti)
                for (int tj = j + 1; tj <= ti && tj < Tm; ++tj) {
                    if (look && ti == j + 1) continue;   // (j + 1, j + 1): wavefront 0's
                    if (t++ % workers != me) continue;
                    tl_trailing_tile(A, j, ti, tj, r16, q);
                }
        }
        __syncthreads();
        if (has_next && !look) {
            if (!tl_diag_wave(A + tl_tile(j + 1, j + 1), min(CH_NB, n - 16 * (j + 1)), Dinv, dinv + 16 * (j + 1), lane) && lane == 0) *s_fail = 1;
            __syncthreads();
        }
        CHPROF(2);
        if (*s_fail) return false;
    }
    return true;
}

// y <- L^-T y with the factor and the block inverses tl_chol left behind.  y: a vector of 16 * tile_rows(n) doubles in LDS whose
// entries >= n are ZERO.  All threads must call.
__device__ __forceinline__ void tl_trsv_t(const double *A, int n, const double *dinv, double *y) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int Tm = tl_tile_rows(n);
    for (int j = Tm - 1; j >= 0; --j) {
        // x_r = sum_{c >= r} Linv[c][r] y_c : the diagonal from dinv, the rest from the upper triangle of the tile, (r, c) = Linv[c][r]
        double xr = 0.0;
        if (tid < 16) {
            const double *tile = A + tl_tile(j, j) + tid;
            const double *yj = y + 16 * j;
            xr = dinv[16 * j + tid] * yj[tid];
#pragma unroll
            for (int c = 1; c < 16; ++c) xr += (c > tid ? tile[c * TL_LD] : 0.0) * yj[c];
        }
        __syncthreads();
        if (tid < 16) y[16 * j + tid] = xr;
        __syncthreads();
        // y_i -= sum_c L[16 j + c][i] x_c for the rows above: thread i reads element (c, i & 15) of tile (j, i >> 4), c = 0 .. 15
        for (int i = tid; i < 16 * j; i += nt) {
            const double *col = A + tl_tile(j, i >> 4) + (i & 15) * TL_LD;
            const double *xj = y + 16 * j;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) s += col[c] * xj[c];
            y[i] -= s;
        }
        __syncthreads();
    }
}

}   // namespace xrhip
