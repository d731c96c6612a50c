// marg_kernels.hip.h -- gfx950 kernels for keyframe marginalisation and IMU pre-integration.
//
// Replaces CeresMarginalizationFactor::marginalize
//   (/root/reference/xrslam/src/xrslam/estimation/ceres/marginalization_factor.h:74-475)
// and PreIntegrator::integrate / increment / compute_sqrt_inv_cov
//   (/root/reference/xrslam/src/xrslam/estimation/preintegrator.cpp:22-100).
//
// Marginalisation pipeline (the linearisation and the landmark Schur product reuse the BA
// kernels with the robust loss switched off):
//   kb_lin_* / kb_landmark / kb_assemble / kb_schur_mfma   unreduced H, b and W^T H_ll^-1 W
//   km_permute     landmark-reduced system with the victim frame moved last
//   km_victim      15x15 inverse of the victim block (pivoted Gauss-Jordan in LDS), T2 = H_rv H_vv^-1
//   km_complement  H' = H_rr - T2 H_vr, b' = b_r - T2 b_v, lower triangle mirrored
//   km_jacobi      symmetric eigen-decomposition by parallel one-sided Jacobi (round-robin pairs)
//   km_finish      sqrt_info = diag(sqrt(lambda)) V^T, infovec = diag(1/sqrt(lambda)) V^T b' (lambda <= 1e-8 -> 0)
#pragma once
#include "ba_kernels.hip.h"
#include "dense_lds.hip.h"

namespace xrhip {

// landmark weights for the marginalisation: omega = 1/H_ll (skipped when not finite)
__global__ __launch_bounds__(256) void km_omega(BaDims d, BaPtrs p) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= d.L) return;
    const double inv = 1.0 / p.hll[l];
    p.omega[l] = isfinite(inv) ? inv : 0.0;
}

// Hm [N x N] (N = 15K) = permuted (Hpp - T on the pose dofs); bm = permuted (gp - W^T (gl/hll)).
// wog = W^T (omega gl): the wide column pass of kb_schur_aux (role 1) on its own
__global__ __launch_bounds__(256) void km_wog(BaDims d, BaPtrs p) { solve_aux_block(d, p, aux_quad_blocks_n(d.n, d.L) + (int)blockIdx.x); }

__global__ __launch_bounds__(256) void km_permute(BaDims d, BaPtrs p, int victim, double *__restrict__ Hm,
                                                  double *__restrict__ bm) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int n = d.n;
    if (e >= n * n) return;
    const int a = e / n, b = e - a * n;
    const int fa = a / 15, ka = a - 15 * fa, fb = b / 15, kb = b - 15 * fb;
    const int pfa = fa < victim ? fa : (fa > victim ? fa - 1 : d.F - 1);
    const int pfb = fb < victim ? fb : (fb > victim ? fb - 1 : d.F - 1);
    double v = p.Hpp[e];
    if (ka < 6 && kb < 6) v -= p.T[(size_t)(6 * fa + ka) * d.PF + 6 * fb + kb];
    const int pa = 15 * pfa + ka, pb = 15 * pfb + kb;
    Hm[(size_t)pa * n + pb] = v;
    if (b == 0) bm[pa] = p.gp[a] - (ka < 6 ? wog_at(d, p, 6 * fa + ka) : 0.0);   // wog = W^T (omega gl), from km_wog
}

// In-LDS inverse of a 15x15 matrix by Gauss-Jordan with partial pivoting; any workgroup size >= 64, all threads
// must call.  A is destroyed, I receives the inverse.  Returns false (uniformly) if a zero pivot is met.
__device__ __forceinline__ bool inv15_block(double (*A)[15], double (*I)[15], int *s_piv, int *s_bad) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < 225; e += nt) I[e / 15][e % 15] = (e / 15 == e % 15) ? 1.0 : 0.0;
    if (tid == 0) *s_bad = 0;
    __syncthreads();
    for (int k = 0; k < 15; ++k) {
        if (tid == 0) {
            int pp = k;
            double best = fabs(A[k][k]);
            for (int r = k + 1; r < 15; ++r)
                if (fabs(A[r][k]) > best) {
                    best = fabs(A[r][k]);
                    pp = r;
                }
            *s_piv = pp;
            if (!(best > 0.0)) *s_bad = 1;
        }
        __syncthreads();
        if (*s_bad) return false;
        const int pp = *s_piv;
        if (pp != k && tid < 15) {
            double t = A[k][tid];
            A[k][tid] = A[pp][tid];
            A[pp][tid] = t;
            t = I[k][tid];
            I[k][tid] = I[pp][tid];
            I[pp][tid] = t;
        }
        __syncthreads();
        const double dkk = A[k][k];
        double f[4];   // column k of the rows this thread updates (read before anything changes)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + r * nt;
            f[r] = e < 225 ? A[e / 15][k] : 0.0;
        }
        __syncthreads();
        if (tid < 15) {
            A[k][tid] /= dkk;
            I[k][tid] /= dkk;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + r * nt;
            if (e < 225) {
                const int i = e / 15, j = e - 15 * i;
                if (i != k) {
                    A[i][j] -= f[r] * A[k][j];
                    I[i][j] -= f[r] * I[k][j];
                }
            }
        }
        __syncthreads();
    }
    return true;
}

// One workgroup (256 threads): invert the victim block (last 15x15 of Hm) and form T2 = H_rv * inv (R x 15).
__global__ __launch_bounds__(256) void km_victim(int N, const double *__restrict__ Hm, double *__restrict__ T2,
                                                 int *__restrict__ status) {
    __shared__ double A[15][15], I[15][15];
    __shared__ int s_piv, s_bad;
    const int R = N - 15, tid = threadIdx.x;
    if (tid < 225) A[tid / 15][tid % 15] = Hm[(size_t)(R + tid / 15) * N + R + tid % 15];
    __syncthreads();
    if (!inv15_block(A, I, &s_piv, &s_bad)) {
        if (tid == 0) *status = 1;
        return;
    }
    for (int e = tid; e < R * 15; e += blockDim.x) {
        const int i = e / 15, j = e - 15 * i;
        double s = 0;
        for (int k = 0; k < 15; ++k) s += Hm[(size_t)i * N + R + k] * I[k][j];
        T2[e] = s;
    }
}

// A [R x R] = H_rr - T2 H_vr (lower triangle mirrored: SelfAdjointEigenSolver reads the lower part),
// bp [R] = b_r - T2 b_v.  One thread per element of the lower triangle's bounding square.
__global__ __launch_bounds__(256) void km_complement(int N, const double *__restrict__ Hm, const double *__restrict__ bm,
                                                     const double *__restrict__ T2, double *__restrict__ A,
                                                     double *__restrict__ bp) {
    const int R = N - 15;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= R * R) return;
    const int i = e / R, j = e - i * R;
    if (j <= i) {
        double v = Hm[(size_t)i * N + j];
        for (int k = 0; k < 15; ++k) v -= T2[i * 15 + k] * Hm[(size_t)(R + k) * N + j];
        A[(size_t)i * R + j] = v;
        A[(size_t)j * R + i] = v;
    }
    if (j == 0) {
        double v = bm[i];
        for (int k = 0; k < 15; ++k) v -= T2[i * 15 + k] * bm[R + k];
        bp[i] = v;
    }
}

// Support of the marginal information matrix.  Frames that share neither a landmark, an IMU factor nor the old
// prior with the victim have exactly zero rows/columns in A (no factor touches them), e.g. all velocity/bias
// dofs of frames that are only connected through vision.  Those rows are zero eigen-directions of the
// reference's SelfAdjointEigenSolver (eigenvalue 0 <= 1e-8 -> dropped), so the decomposition is done on the
// compacted matrix and the result is scattered back with zero rows -- identical Lambda = S^T S and eta.
__global__ __launch_bounds__(512) void km_support(int R, const double *__restrict__ A, const double *__restrict__ bp,
                                                  int *__restrict__ sup_idx, int *__restrict__ sup_n,
                                                  double *__restrict__ As, double *__restrict__ bs) {
    // Round 5 (30 -> ~10 us): the row scan as one pass over the R^2 entries with every load independent of the others (a wavefront
    // per row walked its row in dependent steps), the positions by a prefix count per thread instead of one thread walking the flags
    // and writing the index list to global memory entry by entry, the index list kept in LDS for the compaction (two dependent global
    // loads per entry before).  Pure data movement: the compact matrix is the same.
    __shared__ int flags[512], idx[512], total;
    const int tid = threadIdx.x, nt = blockDim.x;
    flags[tid] = 0;
    __syncthreads();
    const int RR = R * R;
    for (int e0 = tid; e0 < RR; e0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nt;
            v[u] = e < RR ? A[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] != 0.0) flags[(e0 + u * nt) / R] = 1;   // (several threads may store the same 1)
    }
    __syncthreads();
    if (tid < R) {
        int c = 0;
        for (int k = 0; k < tid; ++k) c += flags[k];
        if (flags[tid]) {
            idx[c] = tid;
            sup_idx[c] = tid;
        }
        if (tid == R - 1) {
            total = c + flags[tid];
            *sup_n = c + flags[tid];
        }
    }
    __syncthreads();
    const int Rs = total, SS = Rs * Rs;
    for (int e0 = tid; e0 < SS; e0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nt;
            const int i = e < SS ? e / Rs : 0, j = e < SS ? e - i * Rs : 0;
            v[u] = e < SS ? A[(size_t)idx[i] * R + idx[j]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * nt < SS) As[e0 + u * nt] = v[u];
    }
    for (int i = tid; i < Rs; i += nt) bs[i] = bp[idx[i]];
}

// scatter the compact factor back: sqrt_info [R x R] (zero outside the support), infovec [R]
__global__ __launch_bounds__(256) void km_expand(int R, const int *__restrict__ sup_idx, const int *__restrict__ sup_n,
                                                 const double *__restrict__ Ss, const double *__restrict__ ivs,
                                                 double *__restrict__ sqrt_info, double *__restrict__ infovec) {
    const int Rs = *sup_n;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= R * R) return;
    const int i = e / R, j = e - i * R;
    // row i of the full factor = row i of the compact factor (rows >= Rs are zero), columns scattered
    double v = 0.0;
    if (i < Rs) {
        // find j in the support (sup_idx ascending): binary search
        int lo = 0, hi = Rs - 1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const int sj = sup_idx[mid];
            if (sj == j) {
                v = Ss[(size_t)i * Rs + mid];
                break;
            }
            if (sj < j) lo = mid + 1;
            else hi = mid - 1;
        }
    }
    sqrt_info[e] = v;
    if (j == 0) infovec[i] = (i < Rs) ? ivs[i] : 0.0;
}

// Parallel one-sided (Hestenes) Jacobi on the symmetric matrix A: B = A V is driven to orthogonal columns by
// plane rotations applied to column pairs; the R/2 pairs of a round-robin round are disjoint and processed
// concurrently by G-thread groups.  B, V are column-major R x R (LDS when they fit, else L2-resident global).
// On exit column i of V is an eigenvector and lambda_i = v_i . b_i.
__device__ __forceinline__ int jacobi_sweeps(int R, int G, double *B, double *V, int max_sweeps, int *rotated) {
    const int tid = threadIdx.x;
    const int Rp = R + (R & 1);
    const int pairs = Rp / 2;
    const int k = tid / G, g = tid - k * G;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) *rotated = 0;
        __syncthreads();
        for (int r = 0; r < Rp - 1; ++r) {
            int pi = -1, qi = -1;
            if (k < pairs) {
                if (k == 0) {
                    pi = Rp - 1;
                    qi = r;
                } else {
                    pi = (r + k) % (Rp - 1);
                    qi = (r - k + (Rp - 1)) % (Rp - 1);
                }
                if (pi > qi) {
                    const int t = pi;
                    pi = qi;
                    qi = t;
                }
                if (qi >= R) pi = -1;   // dummy player of an odd-sized tournament
            }
            double al = 0, be = 0, ga = 0;
            if (pi >= 0) {
                const double *bp = B + (size_t)pi * R, *bq = B + (size_t)qi * R;
                for (int e = g; e < R; e += G) {
                    const double x = bp[e], y = bq[e];
                    al += x * x;
                    be += y * y;
                    ga += x * y;
                }
            }
            for (int off = 1; off < G; off <<= 1) {   // groups are aligned power-of-two lane ranges
                al += __shfl_xor(al, off);
                be += __shfl_xor(be, off);
                ga += __shfl_xor(ga, off);
            }
            if (pi >= 0 && ga != 0.0 && fabs(ga) > 1e-15 * sqrt(al * be)) {
                const double zeta = (be - al) / (2.0 * ga);
                double t = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                if (zeta < 0) t = -t;
                if (!isfinite(zeta)) t = 0.0;
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                double *bp = B + (size_t)pi * R, *bq = B + (size_t)qi * R;
                double *vp = V + (size_t)pi * R, *vq = V + (size_t)qi * R;
                for (int e = g; e < R; e += G) {
                    const double x = bp[e], y = bq[e];
                    bp[e] = cs * x - sn * y;
                    bq[e] = sn * x + cs * y;
                    const double u = vp[e], w = vq[e];
                    vp[e] = cs * u - sn * w;
                    vq[e] = sn * u + cs * w;
                }
                if (g == 0) *rotated = 1;
            }
            __syncthreads();
        }
        const int any = *rotated;
        __syncthreads();
        if (!any) break;
    }
    return sweep;
}

// Eigen-decomposition of the compacted matrix As (Rs = *sup_n).  use_lds: B and V live in dynamic LDS
// (2 Rs^2 doubles must fit, checked on the device), otherwise in the global buffers Bg, Vg.
// Outputs the compact factor: Ss row i = sqrt(lambda_i) v_i^T, ivs_i = v_i . bs / sqrt(lambda_i), lambda <= 1e-8 dropped.
__global__ __launch_bounds__(1024) void km_jacobi(const int *__restrict__ sup_n, int lds_doubles,
                                                  const double *__restrict__ As, const double *__restrict__ bs,
                                                  double *__restrict__ Bg, double *__restrict__ Vg,
                                                  double *__restrict__ Ss, double *__restrict__ ivs, int max_sweeps,
                                                  int *__restrict__ sweeps_out, const int *__restrict__ sup_idx = nullptr, int Rf = 0,
                                                  double *__restrict__ sqrt_info_full = nullptr, double *__restrict__ infovec_full = nullptr) {
    extern __shared__ double lds[];
    __shared__ int rotated;
    const int R = *sup_n;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (R == 0) return;
    const bool in_lds = 2 * R * R <= lds_doubles;
    double *B = in_lds ? lds : Bg, *V = in_lds ? lds + (size_t)R * R : Vg;
    for (int e = tid; e < R * R; e += nt) {
        const int c = e / R, r = e - c * R;
        B[e] = As[(size_t)r * R + c];   // column-major copy (A symmetric)
        V[e] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    int G = 8;
    while (G > 1 && ((R + 1) / 2) * G > nt) G >>= 1;
    const int sweeps = jacobi_sweeps(R, G, B, V, max_sweeps, &rotated);
    if (tid == 0) *sweeps_out = sweeps;
    // finish: one wavefront per eigenpair
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int i = wave; i < R; i += nw) {
        const double *v = V + (size_t)i * R, *b = B + (size_t)i * R;
        double lam = 0, vb = 0;
        for (int e = lane; e < R; e += 64) {
            lam += v[e] * b[e];
            vb += v[e] * bs[e];
        }
        lam = wave_sum(lam);
        vb = wave_sum(vb);
        const double sl = lam > 1.0e-8 ? sqrt(lam) : 0.0;
        const double sli = lam > 1.0e-8 ? sqrt(1.0 / lam) : 0.0;
        for (int e = lane; e < R; e += 64) Ss[(size_t)i * R + e] = sl * v[e];
        if (lane == 0) ivs[i] = sli * vb;
    }
    // round 5: the expansion into the full-size prior (km_expand's scatter) in the same launch -- this kernel runs once per sequence
    if (sqrt_info_full) {
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < Rf * Rf; e += nt) {
            const int i = e / Rf, j = e - i * Rf;
            double v = 0.0;
            if (i < R) {
                int lo = 0, hi = R - 1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int sj = sup_idx[mid];
                    if (sj == j) {
                        v = Ss[(size_t)i * R + mid];
                        break;
                    }
                    if (sj < j) lo = mid + 1;
                    else hi = mid - 1;
                }
            }
            sqrt_info_full[e] = v;
        }
        for (int i = tid; i < Rf; i += nt) infovec_full[i] = i < R ? ivs[i] : 0.0;
    }
}

// Cholesky fast path of "create marginalization factor" (marginalization_factor.h:440-455).  When every
// eigenvalue of the marginal information matrix A exceeds the reference's 1e-8 floor, no eigen-direction is
// dropped and ANY factor S with S^T S = A carries the same prior: S = L^T (A = L L^T), infovec = L^-1 b'
// reproduce Lambda = S^T S and eta = S^T infovec exactly like diag(sqrt(lambda)) V^T would.  The kernel
// factors A in LDS and bounds lambda_min from above by inverse iteration; the caller falls back to the
// Jacobi eigen-solver (km_jacobi) when the factorisation fails or the bound is not comfortably above the floor.
// Round 5: the guard's inverse is blocked (tri_inverse_blocked, matrix cores) instead of one forward substitution per thread; the kernel
// also takes over what three launches behind it did on 26 of 27 marginalisations -- the gate (status[4], status[5]: does the eigen path
// have to run, and on what support) and the expansion of its own factor into the full-size prior (km_expand's scatter, straight from
// LDS) -- so that in steady state km_jacobi is one empty launch behind this one and nothing else.
__device__ __forceinline__ bool km_chol_body(int R, int lds_doubles, const double *__restrict__ A, const double *__restrict__ bp,
                                             const int *__restrict__ sup_idx, int Rf, double *__restrict__ sqrt_info_full,
                                             double *__restrict__ infovec_full, double *__restrict__ lam_est, int *__restrict__ status,
                                             double *lds, double (*Dblk)[CH_NB + 1], double *scratch, int *fail, int *s_pos) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (R == 0 || 2 * ((R + 1) & ~1) + R * (R + 1) / 2 > lds_doubles) {
        if (tid == 0) status[2] = 1;   // does not fit: take the Jacobi path
        return false;
    }
    double *x = lds, *y = lds + ((R + 1) & ~1), *Lp = y + ((R + 1) & ~1);
    for (int e = tid; e < R * R; e += nt) {
        const int i = e / R, j = e - i * R;
        if (j <= i) Lp[tri_idx(i, j)] = A[e];
    }
    for (int j = tid; j < Rf; j += nt) s_pos[j] = -1;
    __syncthreads();
    for (int m = tid; m < R; m += nt) s_pos[sup_idx[m]] = m;   // column of the full matrix -> column of the compact one
    if (!chol_blocked(Lp, R, R, Dblk, fail)) {
        if (tid == 0) status[2] = 1;
        return false;
    }
    for (int i = tid; i < R; i += nt) y[i] = bp[i];
    __syncthreads();
    trsv_lower(Lp, R, y);
    // the full-size factor: row i = row i of the compact factor L^T (rows >= R are zero), columns scattered over the support
    for (int i = tid; i < Rf; i += nt) infovec_full[i] = i < R ? y[i] : 0.0;
    for (int e = tid; e < Rf * Rf; e += nt) {
        const int i = e / Rf, j = e - i * Rf;
        double v = 0.0;
        if (i < R) {
            const int m = s_pos[j];
            if (m >= i) v = Lp[tri_idx(m, i)];
        }
        sqrt_info_full[e] = v;
    }
    // Guard of the fast path: every eigenvalue must lie above the reference's 1e-8 clamp.  When a second triangle
    // fits in LDS the bound is rigorous and cheap: trace(A^-1) = |L^-1|_F^2 = sum 1/lambda_i >= 1/lambda_min, so
    // lambda_min >= 1 / trace(A^-1).  Otherwise fall back to inverse iteration (1 / |A^-1 x| for unit x is an upper bound of
    // lambda_min that converges to it) and demand a 100x margin.
    double lam = 0;
    bool rigorous = false;
    if (2 * ((R + 1) & ~1) + R * (R + 1) <= lds_doubles) {
        rigorous = true;
        double *X = Lp + R * (R + 1) / 2;   // packed like L
        __syncthreads();
        tri_inverse_blocked(Lp, X, R);
        double tr = 0;
        for (int e = tid; e < R * (R + 1) / 2; e += nt) tr += X[e] * X[e];
        tr = block_sum(tr, scratch);
        lam = 1.0 / tr;
    } else {
        for (int i = tid; i < R; i += nt) x[i] = 1.0 + 0.5 * sin(1.7 * i + 0.3);
        __syncthreads();
        for (int it = 0; it < 12; ++it) {
            double n2 = 0;
            for (int i = tid; i < R; i += nt) n2 += x[i] * x[i];
            n2 = block_sum(n2, scratch);
            const double inv = 1.0 / sqrt(n2);
            for (int i = tid; i < R; i += nt) y[i] = x[i] * inv;
            __syncthreads();
            trsv_lower(Lp, R, y);
            trsv_lower_t(Lp, R, y);
            double m2 = 0;
            for (int i = tid; i < R; i += nt) {
                m2 += y[i] * y[i];
                x[i] = y[i];
            }
            m2 = block_sum(m2, scratch);
            lam = 1.0 / sqrt(m2);
            __syncthreads();
        }
    }
    const bool good = isfinite(lam) && lam > (rigorous ? 1.0e-8 : 1.0e-6);
    if (tid == 0) {
        lam_est[0] = lam;
        status[3] = good ? 0 : 1;
    }
    return good;
}
__global__ __launch_bounds__(512) void km_chol(const int *__restrict__ sup_n, int lds_doubles,
                                               const double *__restrict__ A, const double *__restrict__ bp,
                                               const int *__restrict__ sup_idx, int Rf, double *__restrict__ sqrt_info_full,
                                               double *__restrict__ infovec_full, double *__restrict__ lam_est,
                                               int *__restrict__ status) {
    extern __shared__ double lds[];
    __shared__ double Dblk[CH_NB][CH_NB + 1];
    __shared__ double scratch[8];
    __shared__ int fail, s_pos[512];
    const int R = *sup_n;
    const bool fast = km_chol_body(R, lds_doubles, A, bp, sup_idx, Rf, sqrt_info_full, infovec_full, lam_est, status, lds, Dblk, scratch,
                                   &fail, s_pos);
    // the gate (round 2's kx_marg_gate): the eigen path runs on the support iff the fast path does not stand
    if (threadIdx.x == 0) {
        status[4] = fast ? 0 : 1;
        status[5] = fast ? 0 : R;
    }
}

// ------------------------------------------------------------------------------------------------------
// IMU pre-integration (PreIntegrator::integrate, estimation/preintegrator.cpp:7-100).  One workgroup of four wavefronts per
// integration; blockIdx.x selects the job.  Euler integration is a recurrence over the samples, but only two short chains of it
// are: per chunk of PI_CHUNK samples
//   P1  lane = sample: bias-corrected rates, expmap(w dt) and the right Jacobian Jr
//   P2  one lane: the quaternion chain q_{n+1} = normalize(q_n * expmap_n) interleaved with the position / velocity chain (and the
//       early delta)
//   P3  lane = sample: R_n, R_{n+1} Jr_n
//   P5  covariance and bias Jacobians in CLOSED FORM over the samples (below): rows of the samples' terms, then a sum over samples
// followed by the 15x15 inverse + Cholesky of the covariance.
//
// P5.  The reference adds a sample at a time (preintegrator.cpp:22-76): Sigma <- A_n Sigma A_n^T + B_n (Q / h_n) B_n^T and, for the
// bias Jacobians J = [dq_dbg 0; dp_dbg dp_dba; dv_dbg dv_dba], J <- A_n J - B_n, with A_n = [E_n 0 0; -h^2/2 R_n hat(a_n), I, h I;
// -h R_n hat(a_n), 0, I], E_n = expmap(w_n h_n)^T as a matrix, B_n = [h Jr_n, 0; 0, h^2/2 R_n; 0, h R_n].  As a loop that is two
// dependent 9x9 products per sample -- 0.95 us of latency each (in-kernel timers, profiles/r04_preintegrate.md): 10 us of a 28 us
// kernel at eleven samples, 50 of 93 at fifty-five.  Unrolled, Sigma_N = sum_n Phi_n B_n (Q / h_n) B_n^T Phi_n^T and
// J_N = -sum_n Phi_n B_n with Phi_n = A_{N-1} ... A_{n+1} -- and that product has a closed form in quantities the two chains
// produce anyway: with R_{m+1} = R_m expmap(w_m h_m) and R hat(a) R^T = hat(R a),
//     Phi_n = [ R_N^T R_s, 0, 0;  -hat(dp) R_s, I, tau I;  -hat(dv) R_s, 0, I ],   s = n + 1,
//     tau = t_N - t_s,  dv = v_N - v_s,  dp = p_N - p_s - v_s tau
// (the sums over m of h_m R_m a_m and of h_m (v_m - v_s) + h_m^2/2 R_m a_m ARE the velocity and position chains).  So every sample's
// D_n = Phi_n B_n (9x6, five non-zero 3x3 blocks) is independent of the others: one work item per (sample, row), then one per
// entry of Sigma / J summing over the samples.  A job longer than a chunk composes its chunks' maps (A, G, C) like the reference
// composes samples ((A2, G2, C2) o (A1, G1, C1) = (A2 A1, A2 G1 A2^T + G2, A2 C1 + C2), preint_compose).  Same mathematics, another
// order of operations: the record agrees with the reference's loop to rounding (tests/test_ba_gpu.py::test_preintegration_parity:
// Jacobians 1e-10, sqrt_inv_cov 1e-7 -- unchanged tolerances; the oracle keeps the reference's loop).  The delta (dt, dq, dp, dv) is
// still the reference's chain, operation for operation.
// Samples and results live in pinned host memory mapped into the device (zero-copy): an integration is a handful of doubles, so a
// copy engine round trip would cost more than the kernel.
struct PreintJob {
    int sample_begin, sample_count;   // into samples [.][7] = t, w, a
    double t_end;
    double bg[3], ba[3];
    int bias_frame, pad_;             // >= 0: the biases are those of this frame in `state_dev` ([.][16], bg at 10, ba at 13) instead
};

constexpr int PI_CHUNK = 32;
constexpr int PI_NT = 256;        // four wavefronts
constexpr int PI_TREE_NT = 192;   // preint_compose runs on the first three
constexpr int PI_HEAD = 2;        // jobs carried in the argument block (a batch is one or two jobs, bar the initialiser's)
constexpr int PI_NE = 111;        // per-chunk sums: 45 (gyroscope part of Sigma, lower triangle) + 21 (accelerometer part: p, v rows) + 45 (J)
// a map (doubles): E, P, V (3x3 row-major), t, C = the five 3x3 blocks dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba, G (9x9)
constexpr int PN_E = 0, PN_P = 9, PN_V = 18, PN_T = 27, PN_C = 28, PN_G = 73, PN_SIZE = 154;

// completion mailbox: the job's status word (pinned host memory) is its last store -- 1 = record complete,
// 3 = covariance not positive definite; the host spins on it instead of synchronising the stream
// (round 6: the record is written with system-scope stores and the wavefronts wait for their acknowledgement -- no write-back and
// invalidation of the whole L2 in front of the status word, host_mailbox.hip.h)
__device__ __forceinline__ void preint_publish(int *status, int code) {
    host_stores_wait();
    __syncthreads();
    if (threadIdx.x == 0) host_store(status + blockIdx.x, code);
}

// out[k] = in[2k+1] o in[2k] for the pairs of a list of maps (an odd last one is carried over); the kernel composes a chunk's map with
// the running one (m = 2).  Work items are ordered by kind, so a wavefront rarely holds two kinds.  Two barriers; every thread calls.
__device__ __forceinline__ void preint_compose(const double *__restrict__ in, int m, double *__restrict__ out,
                                                  double *__restrict__ Tt, bool want_jac, bool want_cov, int tid) {
    const int pairs = m >> 1;
    // ---- stage 1: T = A2 G1 (81 per pair) | E, P, V (27 per pair) | t | the carried-over node
    const int nT = want_cov ? 81 * pairs : 0, nA = 27 * pairs;
    if (tid < PI_TREE_NT) {
        for (int it = tid; it < nT + nA + pairs; it += PI_TREE_NT) {
            if (it < nT) {
                const int k = it / 81, e = it - 81 * k, i = e / 9, j = e - 9 * i, blk = i / 3;
                const double *n1 = in + (size_t)(2 * k) * PN_SIZE, *n2 = n1 + PN_SIZE;
                const double *a = n2 + 9 * blk + 3 * (i - 3 * blk), *G1 = n1 + PN_G;   // row i of A2's first block column
                double v = (a[0] * G1[j] + a[1] * G1[9 + j]) + a[2] * G1[18 + j];
                if (blk == 1) v = (v + G1[9 * i + j]) + n2[PN_T] * G1[9 * (i + 3) + j];
                else if (blk == 2) v = v + G1[9 * i + j];
                Tt[81 * k + e] = v;
            } else if (it < nT + nA) {
                const int q = it - nT, k = q / 27, e = q - 27 * k, blk = e / 9, rc = e - 9 * blk, r = rc / 3, c = rc - 3 * r;
                const double *n1 = in + (size_t)(2 * k) * PN_SIZE, *n2 = n1 + PN_SIZE;
                const double *a = n2 + 9 * blk + 3 * r, *E1 = n1 + PN_E;
                double v = (a[0] * E1[c] + a[1] * E1[3 + c]) + a[2] * E1[6 + c];
                if (blk == 1) v = (v + n1[PN_P + rc]) + n2[PN_T] * n1[PN_V + rc];
                else if (blk == 2) v = v + n1[PN_V + rc];
                out[(size_t)k * PN_SIZE + e] = v;
            } else {
                const int k = it - nT - nA;
                const double *n1 = in + (size_t)(2 * k) * PN_SIZE;
                out[(size_t)k * PN_SIZE + PN_T] = n1[PN_T] + n1[PN_SIZE + PN_T];
            }
        }
        if (m & 1)
            for (int e = tid; e < PN_SIZE; e += PI_TREE_NT) out[(size_t)pairs * PN_SIZE + e] = in[(size_t)(m - 1) * PN_SIZE + e];
    }
    __syncthreads();
    // ---- stage 2: G = T A2^T + G2 (lower triangle, mirrored: 45 per pair) | C = A2 C1 + C2 (45 per pair)
    const int nG = want_cov ? 45 * pairs : 0, nC = want_jac ? 45 * pairs : 0;
    if (tid < PI_TREE_NT) {
        for (int it = tid; it < nG + nC; it += PI_TREE_NT) {
            if (it < nG) {
                const int k = it / 45;
                int li = 0, lj = it - 45 * k;
                while (lj > li) {
                    lj -= li + 1;
                    ++li;
                }
                const double *n2 = in + (size_t)(2 * k + 1) * PN_SIZE, *T = Tt + 81 * k + 9 * li;
                const int blk = lj / 3;
                const double *a = n2 + 9 * blk + 3 * (lj - 3 * blk);   // row lj of A2's first block column
                double v = (T[0] * a[0] + T[1] * a[1]) + T[2] * a[2];
                if (blk == 1) v = (v + T[lj]) + T[lj + 3] * n2[PN_T];
                else if (blk == 2) v = v + T[lj];
                v += n2[PN_G + 9 * li + lj];
                double *Go = out + (size_t)k * PN_SIZE + PN_G;
                Go[9 * li + lj] = v;
                Go[9 * lj + li] = v;
            } else {
                const int q = it - nG, k = q / 45, e = q - 45 * k, blk = e / 9, rc = e - 9 * blk, r = rc / 3, c = rc - 3 * r;
                const double *n1 = in + (size_t)(2 * k) * PN_SIZE, *n2 = n1 + PN_SIZE;
                const double *C1 = n1 + PN_C, *C2 = n2 + PN_C;
                const double t2 = n2[PN_T];
                double v;
                if (blk == 0) {          // dq_dbg = E2 dq_dbg + .
                    const double *a = n2 + PN_E + 3 * r;
                    v = ((a[0] * C1[c] + a[1] * C1[3 + c]) + a[2] * C1[6 + c]) + C2[rc];
                } else if (blk == 1) {   // dp_dbg = P2 dq_dbg + dp_dbg + t2 dv_dbg + .
                    const double *a = n2 + PN_P + 3 * r;
                    v = ((((a[0] * C1[c] + a[1] * C1[3 + c]) + a[2] * C1[6 + c]) + C1[9 + rc]) + t2 * C1[27 + rc]) + C2[9 + rc];
                } else if (blk == 2) {   // dp_dba = dp_dba + t2 dv_dba + .
                    v = (C1[18 + rc] + t2 * C1[36 + rc]) + C2[18 + rc];
                } else if (blk == 3) {   // dv_dbg = V2 dq_dbg + dv_dbg + .
                    const double *a = n2 + PN_V + 3 * r;
                    v = (((a[0] * C1[c] + a[1] * C1[3 + c]) + a[2] * C1[6 + c]) + C1[27 + rc]) + C2[27 + rc];
                } else {                 // dv_dba = dv_dba + .
                    v = C1[36 + rc] + C2[36 + rc];
                }
                out[(size_t)k * PN_SIZE + PN_C + e] = v;
            }
        }
    }
    __syncthreads();
}

// x / d as the compiler's expansion of an f64 division computes it when no operand needs rescaling (|x| <= ~1, d ~ 1: a quaternion
// over its norm): y = the reciprocal refined twice, q = x y, q + (x - d q) y.  div_recip is the part that depends on d alone -- four
// divisions by one norm share it (written once per call below; the compiler merges the identical expressions).
__device__ __forceinline__ double div_recip(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    return y;
}
__device__ __forceinline__ double div_by(double x, double d, double y) {
    const double q = x * y;
    return __builtin_fma(__builtin_fma(-d, q, x), y, q);
}

// blockIdx.x = job of an entry, blockIdx.z = entry: a launch carries the batches of up to XB contexts (group.hip.h)
struct PreintArgs {
    const PreintJob *jobs;
    const double *samples;
    const double *noise_host;
    int want_jac, want_cov;
    double *out;
    int *status;
    int *early;   // per job: 1 once the delta part of its record (doubles 0..10) is in `out` (the full record follows; `status` says when)
    const double *state_dev;
    int n_jobs;
    // the first jobs of the batch again, by value: they arrive with the kernel arguments, so their samples can be requested at once
    // instead of behind a read of `jobs` over the host link (two dependent ~2 us round trips became one)
    PreintJob head[PI_HEAD];
};
__global__ __launch_bounds__(PI_NT) void kp_preintegrate(Batch<PreintArgs> batch) {
    const PreintArgs &ea = batch.e[blockIdx.z];
    if ((int)blockIdx.x >= ea.n_jobs) return;
    const PreintJob *__restrict__ jobs = ea.jobs;
    const double *__restrict__ samples = ea.samples;
    const double *__restrict__ noise_host = ea.noise_host;
    const int want_jac = ea.want_jac, want_cov = ea.want_cov;
    const bool cj = want_cov || want_jac;
    double *__restrict__ out = ea.out;
    int *__restrict__ status = ea.status;
    const double *__restrict__ state_dev = ea.state_dev;
    __shared__ double inv[15][15], sWalk[18];
    __shared__ double maps[2][2 * PN_SIZE], Tt[81];   // [buffer][running map | this chunk's map]
    __shared__ double sJr[PI_CHUNK][9], sRn[PI_CHUNK][9], sM[PI_CHUNK][9];
    __shared__ double sDg[PI_CHUNK][27], sUg[PI_CHUNK][27], sDa[PI_CHUNK][18], sUa[PI_CHUNK][18];
    __shared__ double sPart[2][PI_NE];
    __shared__ double sEq[PI_CHUNK][4], sQ[PI_CHUNK][4], sAc[PI_CHUNK][3], sDt[PI_CHUNK];
    __shared__ double sVn[PI_CHUNK][3], sPn[PI_CHUNK][3], sTn[PI_CHUNK];   // v, p, t AFTER sample n
    __shared__ double sQa[PI_CHUNK][6];   // (q_n a_n) h_n^2 / 2 | (q_n a_n) h_n
    __shared__ double sq[4], sp3[3], sv3[3], sdt, s0v[3], s0p[3], s0t, sRN[9];
    __shared__ double Dinv[CH_NB][CH_NB + 1];
    __shared__ double noise36[36];   // the inputs live in pinned host memory: fetch each of them exactly once
    __shared__ int s_pd;
    const PreintJob job = (blockIdx.x < PI_HEAD) ? ea.head[blockIdx.x] : jobs[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < 36) noise36[tid] = noise_host[tid];
    double *o = out + (size_t)blockIdx.x * XRHIP_IMU_DIM;
    // the running map starts as the identity: E = I, P = V = 0, t = 0, C = 0, G = 0 (what a job without samples reports)
    for (int e = tid; e < PN_SIZE; e += PI_NT) maps[0][e] = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
    if (tid == 0) {
        sq[0] = sq[1] = sq[2] = 0.0;
        sq[3] = 1.0;
        for (int i = 0; i < 3; ++i) sp3[i] = sv3[i] = 0.0;
        sdt = 0.0;
    }
    // a batch queued behind a solve (xrhip_ba_preintegrate_after_solve) starts from the biases that solve left on the device
    const double *bsrc = (job.bias_frame >= 0 && state_dev) ? state_dev + 16 * (size_t)job.bias_frame + 10 : nullptr;
    const V3 bg = bsrc ? v3(bsrc[0], bsrc[1], bsrc[2]) : v3(job.bg[0], job.bg[1], job.bg[2]);
    const V3 ba = bsrc ? v3(bsrc[3], bsrc[4], bsrc[5]) : v3(job.ba[0], job.ba[1], job.ba[2]);
    double walk = 0.0;   // lanes 46..63: one entry of the two 3x3 bias random-walk blocks
    const int wb = (tid - 46) / 9, wr = (tid - 46) - 9 * wb;
    // this lane's sample of the first chunk (t, w, a, the next sample's t); the next chunk's is fetched a chunk ahead -- a read of
    // pinned host memory is ~2 us, paid once instead of once per chunk
    double cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < min(PI_CHUNK, job.sample_count)) {
        const double *smp = samples + (size_t)(job.sample_begin + tid) * 7;
        for (int i = 0; i < 7; ++i) cur[i] = smp[i];
        cur[7] = (tid + 1 < job.sample_count) ? smp[7] : job.t_end;
    }
    __syncthreads();
#ifdef XRHIP_KPROF_PRINT
    long long kt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kt0 = wall_clock64(), ktl = kt0;
#define PI_T(k)                                \
    do {                                       \
        const long long n_ = wall_clock64();   \
        kt[k] += n_ - ktl;                     \
        ktl = n_;                              \
    } while (0)
#else
#define PI_T(k) \
    do {        \
    } while (0)
#endif
    const double wnoise = (tid >= 46 && tid < 64) ? noise36[18 + 9 * wb + wr] : 0.0;
    const bool several = job.sample_count > PI_CHUNK;   // chunks are composed: their maps need the A part too
    int rb = 0;   // maps[rb][0 ..] is the running map
    for (int n0 = 0; n0 < job.sample_count; n0 += PI_CHUNK) {
        const int nc = min(PI_CHUNK, job.sample_count - n0);
        // ---- P1
        double nxt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tid < PI_CHUNK && n0 + PI_CHUNK + tid < job.sample_count) {
            const int n = n0 + PI_CHUNK + tid;
            const double *smp = samples + (size_t)(job.sample_begin + n) * 7;
            for (int i = 0; i < 7; ++i) nxt[i] = smp[i];
            nxt[7] = (n + 1 < job.sample_count) ? smp[7] : job.t_end;
        }
        if (tid < nc) {
            const double dt = cur[7] - cur[0];
            const V3 w = v3(cur[1], cur[2], cur[3]) - bg;
            const V3 a = v3(cur[4], cur[5], cur[6]) - ba;
            const Q4 e = expmap(w * dt);
            sEq[tid][0] = e.x; sEq[tid][1] = e.y; sEq[tid][2] = e.z; sEq[tid][3] = e.w;
            sAc[tid][0] = a.x; sAc[tid][1] = a.y; sAc[tid][2] = a.z;
            sDt[tid] = dt;
            if (cj) {
                const M3 J = right_jacobian(w * dt);
                for (int i = 0; i < 9; ++i) sJr[tid][i] = J.m[i];
            }
        }
        __syncthreads();
        PI_T(0);
        // ---- P2: the chains.  An f64 operation costs its wavefront ~9 cycles whether or not it depends on the previous one (round 6,
        // tools/issue.hip), so a chain on one lane is as long as the instructions it issues -- ~140 per sample when the quaternion chain,
        // the rotation of the acceleration and the position / velocity recurrences shared one loop (0.59 us per sample).  Only
        // q_{n+1} = normalized(q_n e_n) and p, v are recurrences; q_n a_n hangs off q_n alone.  So: (a) the quaternion chain on one lane,
        // its four divisions by the norm sharing one refined reciprocal (the expansion the compiler emits per division, with its
        // reciprocal iterations done once: same quotients); (b) a lane per sample: q_n a_n and its two scaled forms; (c) the p, v, t
        // recurrences on a lane of the second wavefront, beside the first's rotation matrices (P3).  Each value is produced by the
        // operations, in the order, of the reference's loop (the build has floating-point contraction off): the delta's bits are unchanged.
        if (tid == 0) {
            Q4 q = Q4{sq[0], sq[1], sq[2], sq[3]};
            for (int n = 0; n < nc; ++n) {
                sQ[n][0] = q.x; sQ[n][1] = q.y; sQ[n][2] = q.z; sQ[n][3] = q.w;
                const Q4 m = q_mul(q, Q4{sEq[n][0], sEq[n][1], sEq[n][2], sEq[n][3]});
                const double nn = sqrt(m.x * m.x + m.y * m.y + m.z * m.z + m.w * m.w);
                q = Q4{div_by(m.x, nn, div_recip(nn)), div_by(m.y, nn, div_recip(nn)), div_by(m.z, nn, div_recip(nn)),
                       div_by(m.w, nn, div_recip(nn))};
            }
            sq[0] = q.x; sq[1] = q.y; sq[2] = q.z; sq[3] = q.w;
        }
        __syncthreads();
        PI_T(1);
        if (tid < nc) {
            const double h = sDt[tid];
            const V3 qa = q_rot(Q4{sQ[tid][0], sQ[tid][1], sQ[tid][2], sQ[tid][3]}, v3(sAc[tid][0], sAc[tid][1], sAc[tid][2]));
            const V3 a2 = qa * (0.5 * h * h), a1 = qa * h;
            sQa[tid][0] = a2.x; sQa[tid][1] = a2.y; sQa[tid][2] = a2.z;
            sQa[tid][3] = a1.x; sQa[tid][4] = a1.y; sQa[tid][5] = a1.z;
        }
        __syncthreads();
        if (tid == 64) {
            V3 pv = v3(sp3[0], sp3[1], sp3[2]), vv = v3(sv3[0], sv3[1], sv3[2]);
            double T = sdt;
            s0p[0] = pv.x; s0p[1] = pv.y; s0p[2] = pv.z;
            s0v[0] = vv.x; s0v[1] = vv.y; s0v[2] = vv.z;
            s0t = T;
            for (int n = 0; n < nc; ++n) {
                const double h = sDt[n];
                pv = pv + vv * h + v3(sQa[n][0], sQa[n][1], sQa[n][2]);
                vv = vv + v3(sQa[n][3], sQa[n][4], sQa[n][5]);
                T = T + h;
                sPn[n][0] = pv.x; sPn[n][1] = pv.y; sPn[n][2] = pv.z;
                sVn[n][0] = vv.x; sVn[n][1] = vv.y; sVn[n][2] = vv.z;
                sTn[n] = T;
            }
            sp3[0] = pv.x; sp3[1] = pv.y; sp3[2] = pv.z;
            sv3[0] = vv.x; sv3[1] = vv.y; sv3[2] = vv.z;
            sdt = T;
            // The delta (dt, dq, dp, dv) is final with the last chunk's chains -- the covariance, the bias Jacobians and the 15x15
            // factorisation below do not touch it.  It is published now, with a mailbox of its own: the feature tracker reads the
            // delta of the interval the backend's integration covers (same samples, same biases) as soon as it exists, instead of
            // integrating the interval a second time without Jacobians (xrhip_ba_preintegrate_early).  Same values as at the end.
            if (n0 + PI_CHUNK >= job.sample_count) {
                host_store(o + 0, T);
                for (int i = 0; i < 4; ++i) host_store(o + 1 + i, sq[i]);
                host_store(o + 5, pv.x); host_store(o + 6, pv.y); host_store(o + 7, pv.z);
                host_store(o + 8, vv.x); host_store(o + 9, vv.y); host_store(o + 10, vv.z);
                host_stores_wait();
                host_store(ea.early + blockIdx.x, 1);
            }
        }
        // ---- P3 (beside the recurrences): lane = sample: R_n and M_n = R_{n+1} Jr_n; one more lane: R_N of this chunk
        if (cj) {
            if (tid < nc) {
                const M3 R = q_mat(Q4{sQ[tid][0], sQ[tid][1], sQ[tid][2], sQ[tid][3]});
                for (int i = 0; i < 9; ++i) sRn[tid][i] = R.m[i];
                const double *qn = (tid + 1 < nc) ? sQ[tid + 1] : sq;
                const M3 R1 = q_mat(Q4{qn[0], qn[1], qn[2], qn[3]});
                M3 J;
                for (int i = 0; i < 9; ++i) J.m[i] = sJr[tid][i];
                const M3 M = R1 * J;
                for (int i = 0; i < 9; ++i) sM[tid][i] = M.m[i];
            } else if (tid == 128) {
                const M3 R = q_mat(Q4{sq[0], sq[1], sq[2], sq[3]});
                for (int i = 0; i < 9; ++i) sRN[i] = R.m[i];
            }
        }
        if (want_cov && tid >= 46 && tid < 64)
            for (int n = 0; n < nc; ++n) walk += wnoise * sDt[n];
        __syncthreads();
        PI_T(3);
        if (cj) {
            double *chunk = (n0 == 0) ? maps[rb] : maps[rb] + PN_SIZE;   // the first chunk's map IS the running map
            const V3 vN = v3(sv3[0], sv3[1], sv3[2]), pN = v3(sp3[0], sp3[1], sp3[2]);
            const double tN = sdt;
            // ---- P5a: row i of D_n = Phi_n B_n and of U_n = D_n (Q / h_n); rows 0..8 the gyroscope columns, 9..14 the accelerometer's
            for (int it = tid; it < 15 * nc; it += PI_NT) {
                const int n = it / 15, i = it - 15 * n;
                const double h = sDt[n], inv_dt = 1.0 / fmax(h, 1.0e-7);
                const double tau = tN - sTn[n];
                if (i < 9) {
                    const double *M = sM[n];
                    double d0, d1, d2;
                    if (i < 3) {
                        d0 = (sRN[i] * M[0] + sRN[3 + i] * M[3]) + sRN[6 + i] * M[6];
                        d1 = (sRN[i] * M[1] + sRN[3 + i] * M[4]) + sRN[6 + i] * M[7];
                        d2 = (sRN[i] * M[2] + sRN[3 + i] * M[5]) + sRN[6 + i] * M[8];
                        d0 = h * d0;
                        d1 = h * d1;
                        d2 = h * d2;
                    } else {
                        const V3 vs = v3(sVn[n][0], sVn[n][1], sVn[n][2]);
                        V3 x = vN - vs;
                        if (i < 6) x = (pN - v3(sPn[n][0], sPn[n][1], sPn[n][2])) - vs * tau;
                        const int r = (i < 6) ? i - 3 : i - 6;
                        // row r of hat(x): the two other components with signs; (hat(x) M)[r][c] = xa * M[ka][c] + xb * M[kb][c]
                        const int ka = (r + 1) % 3, kb = (r + 2) % 3;
                        const double xa = -get(x, kb), xb = get(x, ka);
                        d0 = -h * (xa * M[3 * ka] + xb * M[3 * kb]);
                        d1 = -h * (xa * M[3 * ka + 1] + xb * M[3 * kb + 1]);
                        d2 = -h * (xa * M[3 * ka + 2] + xb * M[3 * kb + 2]);
                    }
                    double *D = sDg[n] + 3 * i, *U = sUg[n] + 3 * i;
                    D[0] = d0; D[1] = d1; D[2] = d2;
                    for (int k = 0; k < 3; ++k)
                        U[k] = (d0 * (noise36[k] * inv_dt) + d1 * (noise36[3 + k] * inv_dt)) + d2 * (noise36[6 + k] * inv_dt);
                } else {
                    const int r = i - 9, rr = (r < 3) ? r : r - 3;
                    const double coef = (r < 3) ? (0.5 * h * h + tau * h) : h;
                    const double *R = sRn[n] + 3 * rr;
                    const double d0 = coef * R[0], d1 = coef * R[1], d2 = coef * R[2];
                    double *D = sDa[n] + 3 * r, *U = sUa[n] + 3 * r;
                    D[0] = d0; D[1] = d1; D[2] = d2;
                    for (int k = 0; k < 3; ++k)
                        U[k] = (d0 * (noise36[9 + k] * inv_dt) + d1 * (noise36[12 + k] * inv_dt)) + d2 * (noise36[15 + k] * inv_dt);
                }
            }
            __syncthreads();
            // ---- P5b: the sums over the samples, two halves per entry
            const int half = (nc + 1) >> 1;
            if (tid < 2 * PI_NE) {
                const int part = tid / PI_NE, e = tid - PI_NE * part;
                const int nb = part ? half : 0, ne = part ? nc : half;
                double acc = 0.0;
                if (e < 66) {
                    if (want_cov) {
                        const bool gy = e < 45;
                        int li = 0, lj = gy ? e : e - 45;
                        while (lj > li) {
                            lj -= li + 1;
                            ++li;
                        }
                        for (int n = nb; n < ne; ++n) {
                            const double *U = (gy ? sUg[n] : sUa[n]) + 3 * li, *D = (gy ? sDg[n] : sDa[n]) + 3 * lj;
                            acc += (U[0] * D[0] + U[1] * D[1]) + U[2] * D[2];
                        }
                    }
                } else if (want_jac) {
                    const int q = e - 66, blk = q / 9, rc = q - 9 * blk;   // dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba
                    const bool gy = blk != 2 && blk != 4;
                    const int off = (blk == 0) ? rc : (blk == 1) ? 9 + rc : (blk == 2) ? rc : (blk == 3) ? 18 + rc : 9 + rc;
                    for (int n = nb; n < ne; ++n) acc += gy ? sDg[n][off] : sDa[n][off];
                }
                sPart[part][e] = acc;
            }
            __syncthreads();
            // ---- P5c: this chunk's map
            if (tid < 81) {
                const int i = tid / 9, j = tid - 9 * i, li = max(i, j), lj = min(i, j);
                double g = sPart[0][li * (li + 1) / 2 + lj] + sPart[1][li * (li + 1) / 2 + lj];
                if (lj >= 3) {
                    const int ai = li - 3, aj = lj - 3;
                    g += sPart[0][45 + ai * (ai + 1) / 2 + aj] + sPart[1][45 + ai * (ai + 1) / 2 + aj];
                }
                chunk[PN_G + tid] = g;
            } else if (tid < 81 + 45) {
                const int e = tid - 81;
                chunk[PN_C + e] = -(sPart[0][66 + e] + sPart[1][66 + e]);
            } else if (tid < 81 + 45 + 27 && several) {
                // A part = Phi at the chunk's first state: [R_N^T R_0, 0, 0; -hat(dp) R_0, I, tau I; -hat(dv) R_0, 0, I]
                const int e = tid - 126, blk = e / 9, rc = e - 9 * blk, r = rc / 3, c = rc - 3 * r;
                const double *R0 = sRn[0];
                const V3 v0 = v3(s0v[0], s0v[1], s0v[2]);
                const double tau = tN - s0t;
                double v;
                if (blk == 0) {
                    v = (sRN[r] * R0[c] + sRN[3 + r] * R0[3 + c]) + sRN[6 + r] * R0[6 + c];
                } else {
                    V3 x = vN - v0;
                    if (blk == 1) x = (pN - v3(s0p[0], s0p[1], s0p[2])) - v0 * tau;
                    const int ka = (r + 1) % 3, kb = (r + 2) % 3;
                    v = -(-get(x, kb) * R0[3 * ka + c] + get(x, ka) * R0[3 * kb + c]);
                }
                chunk[e] = v;
            } else if (tid == 81 + 45 + 27 && several) {
                chunk[PN_T] = tN - s0t;
            }
            __syncthreads();
            if (n0 > 0) {   // running <- chunk o running
                preint_compose(maps[rb], 2, maps[rb ^ 1], Tt, want_jac, want_cov, tid);
                rb ^= 1;
            }
        }
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
        PI_T(4);
    }
    const double *res = maps[rb];
    // outputs
    if (tid == 0) {
        host_store(o + 0, sdt);
        for (int i = 0; i < 4; ++i) host_store(o + 1 + i, sq[i]);
        for (int i = 0; i < 3; ++i) {
            host_store(o + 5 + i, sp3[i]);
            host_store(o + 8 + i, sv3[i]);
        }
        if (job.sample_count <= 0) {   // no chunk ran, so nobody published the early delta: it is the identity just written
            host_stores_wait();
            host_store(ea.early + blockIdx.x, 1);
        }
    }
    if (tid < 45) host_store(o + 11 + tid, want_jac ? res[PN_C + tid] : 0.0);
    if (!want_cov) {
        for (int e = tid; e < 225; e += PI_NT) host_store(o + 56 + e, 0.0);
        preint_publish(status, 1);
        return;
    }
    if (tid >= 46 && tid < 64) sWalk[tid - 46] = walk;
    // sqrt_inv_cov = LLT(cov^-1).matrixL().transpose().  With J the index reversal and J cov J = Lr Lr^T,
    // cov^-1 = (J Lr^-T J)(J Lr^-1 J) and J Lr^-T J is lower triangular with a positive diagonal, i.e. it IS that
    // Cholesky factor: sqrt_inv_cov = J Lr^-1 J.  One register-resident 15x15 factorisation + triangular inverse
    // (chol_diag_wave) instead of a pivoted Gauss-Jordan inverse followed by a Cholesky.
    __syncthreads();
    double *Pr = &inv[0][0];   // packed lower triangle of J cov J; cov = [G 0; 0 blockdiag(walk_bg, walk_ba)]
    for (int e = tid; e < 120; e += PI_NT) {
        int i = 0, j = e;
        while (j > i) {
            j -= i + 1;
            ++i;
        }
        const int a = 14 - i, b = 14 - j;   // a <= b
        double v = 0.0;
        if (b < 9) v = res[PN_G + 9 * a + b];
        else if (a >= 9 && (a - 9) / 3 == (b - 9) / 3) v = sWalk[9 * ((a - 9) / 3) + 3 * ((a - 9) % 3) + (b - 9) % 3];
        Pr[tri_idx(i, j)] = v;
    }
    __syncthreads();
    PI_T(5);
    if (tid < 64) {
        const bool pd = chol_diag_wave(Pr, 0, 15, Dinv, tid);
        if (tid == 0) s_pd = pd ? 1 : 0;
    }
    __syncthreads();
    PI_T(6);
    if (!s_pd) {
        preint_publish(status, 3);
        return;
    }
    for (int e = tid; e < 225; e += PI_NT) {
        const int i = e / 15, j = e - 15 * i;
        host_store(o + 56 + e, (j >= i) ? Dinv[14 - i][14 - j] : 0.0);   // upper triangular, row-major
    }
    preint_publish(status, 1);
#ifdef XRHIP_KPROF_PRINT
    PI_T(7);
    if (tid == 0 && blockIdx.x == 0 && blockIdx.z == 0)
        printf("kp_preintegrate samples %d jac %d: P1 %lld chains %lld P3 %lld - %lld P5 %lld pack %lld chol %lld out %lld total %lld x10ns\n",
               job.sample_count, want_jac, kt[0], kt[1], kt[2], kt[3], kt[4], kt[5], kt[6], kt[7], wall_clock64() - kt0);
#endif
}

}   // namespace xrhip
