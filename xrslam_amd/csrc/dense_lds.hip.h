// dense_lds.hip.h -- workgroup-cooperative dense SPD kernels on a packed lower-triangular matrix that
// lives in LDS (or, for systems that exceed the 160 KB LDS of a CU, in an L2-resident global buffer).
//
// Used by the reduced-camera-system solve of the bundle adjustment (kb_solve) and by the Cholesky
// fast path of the marginalisation (km_chol).  Right-looking blocked Cholesky with 16-wide panels:
//   (1) the 16x16 diagonal block is factored by wavefront 0 alone (lane-parallel, wave-synchronous,
//       no workgroup barrier inside),
//   (2) the panel below it is solved one row per thread,
//   (3) the trailing matrix gets its rank-16 update from the f64 matrix cores (16x16 tiles, one per wavefront),
// i.e. 3 workgroup barriers per panel instead of 3 per column.  Triangular solves are blocked the same way.
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

// In-place blocked Cholesky of the packed lower triangle A (n x n): on success A holds L.
// D is a [CH_NB][CH_NB+1] LDS scratch block, s_fail an LDS flag.  All threads of the workgroup must call.
// Returns false (uniformly) if a non-positive pivot is met.
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
__device__ __forceinline__ bool chol_blocked(double *A, int n, double (*D)[CH_NB + 1], int *s_fail,
                                             long long *prof = nullptr) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
#ifdef XRHIP_KPROF
    long long t_prev = wall_clock64();
#endif
    if (tid == 0) *s_fail = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = min(CH_NB, n - j0);
        // ---- (1) diagonal block, wavefront 0
        if (wave == 0) {
            for (int p = lane; p < CH_NB * CH_NB; p += 64) {
                const int r = p >> 4, k = p & 15;
                if (r < nb && k <= r) D[r][k] = A[tri_idx(j0 + r, j0 + k)];
            }
            wave_sync();
            for (int c = 0; c < nb; ++c) {
                const double dcc = D[c][c];
                if (!(dcc > 0.0) || !isfinite(dcc)) {
                    if (lane == 0) *s_fail = 1;
                    break;
                }
                const double dd = sqrt(dcc);
                if (lane > c && lane < nb) D[lane][c] = D[lane][c] / dd;
                if (lane == c) D[c][c] = dd;
                wave_sync();
                for (int p = lane; p < CH_NB * CH_NB; p += 64) {
                    const int r = p >> 4, k = p & 15;
                    if (k > c && r >= k && r < nb) D[r][k] -= D[r][c] * D[k][c];
                }
                wave_sync();
            }
            for (int p = lane; p < CH_NB * CH_NB; p += 64) {
                const int r = p >> 4, k = p & 15;
                if (r < nb && k <= r) A[tri_idx(j0 + r, j0 + k)] = D[r][k];
            }
        }
        __syncthreads();
        CHPROF(0);
        if (*s_fail) return false;
        const int jb = j0 + nb;
        if (jb >= n) break;
        // ---- (2) panel solve: row i of the panel, x = A[i][j0..j0+nb) * Ldd^-T
        for (int i = jb + tid; i < n; i += nt) {
            double x[CH_NB];
            double *row = A + tri_idx(i, j0);
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                if (c < nb) {
                    double s = row[c];
                    for (int k = 0; k < c; ++k) s -= x[k] * D[c][k];
                    x[c] = s / D[c][c];
                }
            }
#pragma unroll
            for (int c = 0; c < CH_NB; ++c)
                if (c < nb) row[c] = x[c];
        }
        __syncthreads();
        CHPROF(1);
        // ---- (3) trailing update A22 -= P P^T (P = the freshly solved n-jb x 16 panel) on the f64 matrix cores:
        // 16x16 output tiles of the lower triangle are dealt round-robin to the wavefronts; one tile = four
        // v_mfma_f64_16x16x4_f64 (k = 16).  Operand layout: A[i][k] from lane (i = lane & 15, k = lane >> 4),
        // B[k][j] from lane (k = lane >> 4, j = lane & 15), D[(lane >> 4) + 4 r][lane & 15] in register r.
        {
            const int nw = nt >> 6, r16 = lane & 15, q = lane >> 4;
            const int T = (n - jb + 15) >> 4;
            int t = 0;
            for (int ti = 0; ti < T; ++ti)
                for (int tj = 0; tj <= ti; ++tj, ++t) {
                    if (t % nw != wave) continue;
                    const int gi = jb + 16 * ti + r16, gk = jb + 16 * tj + r16;
                    const double *pa = A + tri_idx(min(gi, n - 1), j0) + q;
                    const double *pb = A + tri_idx(min(gk, n - 1), j0) + q;
                    const bool va = gi < n, vb = gk < n;
                    typedef double d4 __attribute__((ext_vector_type(4)));
                    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const double av = va ? pa[4 * s4] : 0.0, bv = vb ? pb[4 * s4] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oi = jb + 16 * ti + q + 4 * r, ok = jb + 16 * tj + r16;
                        if (oi < n && ok <= oi) A[tri_idx(oi, ok)] -= acc[r];
                    }
                }
        }
        __syncthreads();
        CHPROF(2);
    }
    return true;
}

// y <- L^-1 y  (forward) with L packed lower in A.  All threads must call.
__device__ __forceinline__ void trsv_lower(const double *A, int n, double *y) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    for (int j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = min(CH_NB, n - j0);
        if (wave == 0) {
            for (int c = 0; c < nb; ++c) {
                const double yc = y[j0 + c] / A[tri_idx(j0 + c, j0 + c)];
                wave_sync();
                if (lane == c) y[j0 + c] = yc;
                if (lane > c && lane < nb) y[j0 + lane] -= A[tri_idx(j0 + lane, j0 + c)] * yc;
                wave_sync();
            }
        }
        __syncthreads();
        const int jb = j0 + nb;
        for (int i = jb + tid; i < n; i += nt) {
            const double *row = A + tri_idx(i, j0);
            double s = 0;
            for (int c = 0; c < nb; ++c) s += row[c] * y[j0 + c];
            y[i] -= s;
        }
        __syncthreads();
    }
}

// y <- L^-T y  (backward).  All threads must call.
__device__ __forceinline__ void trsv_lower_t(const double *A, int n, double *y) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int last = ((n - 1) / CH_NB) * CH_NB;
    for (int j0 = last; j0 >= 0; j0 -= CH_NB) {
        const int nb = min(CH_NB, n - j0);
        if (wave == 0) {
            for (int c = nb - 1; c >= 0; --c) {
                const double yc = y[j0 + c] / A[tri_idx(j0 + c, j0 + c)];
                wave_sync();
                if (lane == c) y[j0 + c] = yc;
                if (lane < c) y[j0 + lane] -= A[tri_idx(j0 + c, j0 + lane)] * yc;
                wave_sync();
            }
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

}   // namespace xrhip
