// Standalone check + timing of the workgroup Cholesky of csrc/dense_lds.hip.h (development aid, GPU box):
//   packed (round 1/2: chol_blocked + trsv_lower_t)  vs  tiled (round 3: tl_chol + tl_trsv_t),
// the system [S | rhs] of a reduced camera system: n unknowns, the right-hand side riding along as row n.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I xrslam_amd/csrc tools/chol_test.hip -o xrslam_amd/bin/xr-chol-test
// Prints, per size: max relative error of the solution of S x = rhs against a host double-double-free reference (plain double
// Cholesky), agreement of the two device variants, and the in-kernel time of factorisation + substitution (100 MHz clock).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef XR_TL_TIMELINE   // -DXR_TL_TIMELINE: tl_chol's phases in shader cycles as wavefront 0 sees them (see dense_lds.hip.h)
__shared__ long long xr_tl_acc[16];
__shared__ long long xr_tl_prev;
#define XR_TL_CLK_RESET() do { if (threadIdx.x < 16) xr_tl_acc[threadIdx.x] = 0; __syncthreads(); if (threadIdx.x < 64) xr_tl_prev = __builtin_readcyclecounter(); } while (0)
#define XR_TL_CLK(slot, on) do { if (on) { const long long n_ = __builtin_readcyclecounter(); xr_tl_acc[slot] += n_ - xr_tl_prev; xr_tl_prev = n_; } } while (0)
#endif
#include "dense_lds.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace xrhip;

// src: packed lower triangle [n(n+1)/2] + rhs [n]; out: solution [n]; ticks[0] = factorisation, [1] = substitution
template <int NT> __global__ __launch_bounds__(NT) void k_packed(const double *src, int n, double *out, long long *ticks, int *failed) {
    extern __shared__ double lds[];
    __shared__ double Dblk[CH_NB][CH_NB + 1];
    __shared__ int fail;
    double *A = lds, *y = A + tri_idx(n, 0);
    const int tri = n * (n + 1) / 2;
    for (int e = threadIdx.x; e < tri + n; e += NT) A[e] = src[e];
    __syncthreads();
    const long long t0 = wall_clock64();
    const bool ok = chol_blocked(A, n, n + 1, Dblk, &fail);
    __syncthreads();
    const long long t1 = wall_clock64();
    if (ok) trsv_lower_t(A, n, y);
    __syncthreads();
    const long long t2 = wall_clock64();
    for (int i = threadIdx.x; i < n; i += NT) out[i] = y[i];
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = t2 - t1;
        *failed = ok ? 0 : 1;
    }
}

template <int NT> __global__ __launch_bounds__(NT) void k_tiled(const double *src, int n, double *out, double *Lout, long long *ticks, int *failed, long long *prof) {
    extern __shared__ double lds[];
    __shared__ double Dblk[CH_NB][CH_NB + 1];
    __shared__ int fail;
    const int T = tl_tile_rows(n + 1);
    double *A = lds, *yv = A + tl_doubles(n + 1);
    const int tri = n * (n + 1) / 2;
    const long long tc0 = wall_clock64();
    tl_clear(A, n, n + 1);
    __syncthreads();
    {   // copy-in by tiles: lane (c = lane & 15, rr = lane >> 4), four rows per pass -> 16 lanes read 128 contiguous bytes
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = NT >> 6, c = lane & 15, rr = lane >> 4;
        const int nt_tiles = T * (T + 1) / 2;
        for (int t = wave; t < nt_tiles; t += nw) {
            int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;
            double v[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int i = 16 * ti + rr + 4 * ps, j = 16 * tj + c;
                v[ps] = (i <= n && j <= i && j < n) ? (i < n ? src[i * (i + 1) / 2 + j] : src[tri + j]) : ((i == j && i >= n) ? 1.0 : 0.0);
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) A[tl_tile(ti, tj) + c * TL_LD + rr + 4 * ps] = v[ps];
        }
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    const bool ok = tl_chol(A, n, n + 1, &Dblk[0][0], &fail, prof);   // prof: phase timers, -DXRHIP_KPROF builds only
    __syncthreads();
    const long long t1 = wall_clock64();
    for (int i = threadIdx.x; i < 16 * T; i += NT) yv[i] = i < n ? A[tl_idx(n, i)] : 0.0;
    __syncthreads();
    if (ok) tl_trsv_t(A, n, yv);
    __syncthreads();
    const long long t2 = wall_clock64();
    for (int i = threadIdx.x; i < n; i += NT) out[i] = yv[i];
    for (int e = threadIdx.x; e < tri; e += NT) {
        int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > e) --i;
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        const int jj = e - i * (i + 1) / 2;
        // below the diagonal tiles: L; inside a diagonal tile: the block's inverse, kept transposed ((r, c) holds Linv[c][r])
        Lout[e] = ((i >> 4) == (jj >> 4)) ? A[tl_tile(i >> 4, i >> 4) + (i & 15) * TL_LD + (jj & 15)] : A[tl_idx(i, jj)];
    }
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = t2 - t1;
        ticks[2] = t0 - tc0;
        *failed = ok ? 0 : 1;
#ifdef XR_TL_TIMELINE
        for (int i = 0; i < 10; ++i) prof[i] = xr_tl_acc[i];
#endif
    }
}

static double urand() { return rand() / (double)RAND_MAX - 0.5; }

template <int NT> static void run(int n) {
    const int tri = n * (n + 1) / 2;
    // SPD, Jacobi-scaled like the reduced camera system: S = D (G G^T + n I) D with unit diagonal
    std::vector<double> G(n * n), S(n * n), src(tri + n), rhs(n);
    for (auto &g : G) g = urand();
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = (i == j) ? 0.05 * n : 0.0;
            for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k];
            S[i * n + j] = S[j * n + i] = s;
        }
    for (int i = 0; i < n; ++i) rhs[i] = urand();
    std::vector<double> dg(n);
    for (int i = 0; i < n; ++i) dg[i] = 1.0 / sqrt(S[i * n + i]);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[i * n + j] *= dg[i] * dg[j];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) src[i * (i + 1) / 2 + j] = S[i * n + j];
    for (int i = 0; i < n; ++i) src[tri + i] = rhs[i];
    // host reference
    std::vector<double> L(S), x(rhs);
    for (int j = 0; j < n; ++j) {
        double d = L[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        d = sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = L[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
    double *d_src, *d_out, *d_L;
    long long *d_t;
    int *d_f;
    CK(hipMalloc(&d_src, sizeof(double) * (tri + n)));
    CK(hipMalloc(&d_out, sizeof(double) * n));
    CK(hipMalloc(&d_L, sizeof(double) * tri));
    CK(hipMalloc(&d_t, sizeof(long long) * 4));
    CK(hipMalloc(&d_f, sizeof(int)));
    CK(hipMemcpy(d_src, src.data(), sizeof(double) * (tri + n), hipMemcpyHostToDevice));
    std::vector<double> xp(n), xt(n), Lt(tri);
    long long tp[4] = {0}, tt[4] = {0}, hp[16] = {0};
    long long *d_p;
    CK(hipMalloc(&d_p, sizeof(hp)));
    int fp = 0, ft = 0;
    const size_t lds_p = sizeof(double) * (size_t)((n + 1) * (n + 2) / 2 + 16);
    const size_t lds_t = sizeof(double) * (size_t)(tl_doubles(n + 1) + 32 * tl_tile_rows(n + 1));
    CK(hipFuncSetAttribute((const void *)k_packed<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    CK(hipFuncSetAttribute((const void *)k_tiled<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_packed<NT>, dim3(1), dim3(NT), lds_p, 0, d_src, n, d_out, d_t, d_f);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(xp.data(), d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tp, d_t, sizeof(long long) * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&fp, d_f, sizeof(int), hipMemcpyDeviceToHost));
        CK(hipMemset(d_p, 0, sizeof(hp)));
        hipLaunchKernelGGL(k_tiled<NT>, dim3(1), dim3(NT), lds_t, 0, d_src, n, d_out, d_L, d_t, d_f, d_p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(xt.data(), d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Lt.data(), d_L, sizeof(double) * tri, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tt, d_t, sizeof(long long) * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ft, d_f, sizeof(int), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hp, d_p, sizeof(hp), hipMemcpyDeviceToHost));
    }
    double xmax = 0, ep = 0, et = 0, el = 0, lmax = 0, ept = 0;
    for (int i = 0; i < n; ++i) xmax = fmax(xmax, fabs(x[i]));
    for (int i = 0; i < n; ++i) {
        ep = fmax(ep, fabs(xp[i] - x[i]));
        et = fmax(et, fabs(xt[i] - x[i]));
        ept = fmax(ept, fabs(xt[i] - xp[i]));
    }
    // the tiled form keeps the inverse of every 16x16 diagonal block of L in that block's place: same for the reference
    std::vector<double> Lr(L);
    for (int b0 = 0; b0 < n; b0 += 16) {
        const int nbk = n - b0 < 16 ? n - b0 : 16;
        for (int m = 0; m < nbk; ++m)
            for (int r = 0; r < nbk; ++r) {
                double sacc = (r == m) ? 1.0 : 0.0;
                for (int k = m; k < r; ++k) sacc -= L[(b0 + r) * n + b0 + k] * Lr[(b0 + k) * n + b0 + m];
                Lr[(b0 + r) * n + b0 + m] = (r < m) ? 0.0 : sacc / L[(b0 + r) * n + b0 + r];
            }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            lmax = fmax(lmax, fabs(Lr[i * n + j]));
            el = fmax(el, fabs(Lt[i * (i + 1) / 2 + j] - Lr[i * n + j]));
        }
    printf("{\"threads\": %d, \"n\": %d, \"packed_fail\": %d, \"tiled_fail\": %d, \"x_err_packed\": %.2e, \"x_err_tiled\": %.2e, \"tiled_vs_packed\": %.2e, "
           "\"L_err_tiled\": %.2e, \"packed_us\": [%.2f, %.2f], \"tiled_us\": [%.2f, %.2f], \"tiled_copy_in_us\": %.2f}\n",
           NT, n, fp, ft, ep / xmax, et / xmax, ept / xmax, el / lmax, tp[0] * 0.01, tp[1] * 0.01, tt[0] * 0.01, tt[1] * 0.01, tt[2] * 0.01);
#ifdef XRHIP_KPROF
    // tl_chol's phases as thread 0 sees them: first block | panels | trailing update + next block (of which: load, factor, write-back)
    printf("{\"kprof_tiled_us\": true, \"threads\": %d, \"n\": %d, \"first_block\": %.2f, \"panels\": %.2f, \"trailing_and_next_block\": %.2f, \"next_block_load\": %.2f, "
           "\"next_block_factor\": %.2f, \"next_block_store\": %.2f}\n", NT, n, hp[0] * 0.01, hp[1] * 0.01, hp[2] * 0.01, hp[4] * 0.01, hp[5] * 0.01, hp[6] * 0.01);
#endif
#ifdef XR_TL_TIMELINE
    printf("{\"tl_chol_wave0_cycles\": true, \"threads\": %d, \"n\": %d, \"first_block\": %lld, \"barrier_a\": %lld, \"panel\": %lld, \"barrier_b\": %lld, \"own_trailing_tile\": %lld, "
           "\"block_load\": %lld, \"block_factor\": %lld, \"block_store\": %lld, \"barrier_c\": %lld, \"others_trailing_as_seen\": %lld}\n", NT, n, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7], hp[8], hp[9]);
#endif
    CK(hipFree(d_p));
    CK(hipFree(d_src)); CK(hipFree(d_out)); CK(hipFree(d_L)); CK(hipFree(d_t)); CK(hipFree(d_f));
}


// ---- the 16x16 diagonal block alone (tl_diag_wave): time per call and the block's inverse against a host Cholesky.
// src: 16x16 symmetric block, row-major; nb pivots (rows >= nb: identity padding, as tl_clear leaves them).
__global__ __launch_bounds__(64) void k_diag(const double *src, int nb, int reps, long long *ticks, double *Iout, int *failed) {
    __shared__ double tile[TL_TILE], keep[TL_TILE], ident[TL_TILE];
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) {
        const int r = e >> 4, c = e & 15;
        keep[c * TL_LD + r] = (r < nb && c < nb) ? src[r * 16 + c] : (r == c ? 1.0 : 0.0);
        ident[c * TL_LD + r] = r == c ? 1.0 : 0.0;
    }
    for (int e = lane; e < TL_TILE; e += 64) tile[e] = 0.0;
    __syncthreads();
    bool ok = true;
    long long t0 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = lane; e < 256; e += 64) tile[(e & 15) * TL_LD + (e >> 4)] = keep[(e & 15) * TL_LD + (e >> 4)];
        wave_sync();
        ok = tl_diag_wave(tile, nb, ident, lane) && ok;
        wave_sync();
    }
    long long t1 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {   // the copy alone
        for (int e = lane; e < 256; e += 64) tile[(e & 15) * TL_LD + (e >> 4)] = keep[(e & 15) * TL_LD + (e >> 4)] + (double)rep * 0.0;
        wave_sync();
    }
    long long t2 = wall_clock64();
    // once more for the outputs (the timing loop above left the copy in `tile`)
    ok = tl_diag_wave(tile, nb, ident, lane) && ok;
    wave_sync();
    for (int e = lane; e < 256; e += 64) {
        const int r = e >> 4, c = e & 15;
        Iout[e] = tile[r * TL_LD + c];   // Linv[r][c] sits at (c, r)
    }
    if (lane == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = t2 - t1;
        *failed = ok ? 0 : 1;
    }
}

static void run_diag(int nb) {
    double S[256], L[256] = {0}, Li[256] = {0};
    {
        double G[256];
        for (auto &g : G) g = urand();
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = (i == j) ? 0.8 : 0.0;
                for (int k = 0; k < 16; ++k) s += G[i * 16 + k] * G[j * 16 + k];
                S[i * 16 + j] = s;
            }
        double dg[16];
        for (int i = 0; i < 16; ++i) dg[i] = 1.0 / sqrt(S[i * 16 + i]);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) S[i * 16 + j] *= dg[i] * dg[j];
    }
    // host: Cholesky of the leading nb x nb block, identity below; its inverse
    for (int j = 0; j < 16; ++j) {
        if (j >= nb) { L[j * 16 + j] = 1.0; continue; }
        double d = S[j * 16 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 16 + k] * L[j * 16 + k];
        L[j * 16 + j] = sqrt(d);
        for (int i = j + 1; i < nb; ++i) {
            double s = S[i * 16 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 16 + k] * L[j * 16 + k];
            L[i * 16 + j] = s / L[j * 16 + j];
        }
    }
    for (int m = 0; m < 16; ++m)
        for (int r = 0; r < 16; ++r) {
            double s = (r == m) ? 1.0 : 0.0;
            for (int k = 0; k < r; ++k) s -= L[r * 16 + k] * Li[k * 16 + m];
            Li[r * 16 + m] = s / L[r * 16 + r];
        }
    double *d_src, *d_I;
    long long *d_t;
    int *d_f;
    CK(hipMalloc(&d_src, sizeof(S)));
    CK(hipMalloc(&d_I, sizeof(S)));
    CK(hipMalloc(&d_t, sizeof(long long) * 16));
    CK(hipMalloc(&d_f, sizeof(int)));
    CK(hipMemcpy(d_src, S, sizeof(S), hipMemcpyHostToDevice));
    const int reps = 200;
    long long t[16] = {0};
    double gI[256];
    int f = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_diag, dim3(1), dim3(64), 0, 0, d_src, nb, reps, d_t, d_I, d_f);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(t, d_t, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gI, d_I, sizeof(S), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&f, d_f, sizeof(int), hipMemcpyDeviceToHost));
    double eI = 0, mI = 0;
    for (int i = 0; i < nb; ++i)   // the pivot rows' inverse (rows >= nb of the tile keep their own entries)
        for (int j = 0; j < nb; ++j) {
            eI = fmax(eI, fabs(gI[i * 16 + j] - Li[i * 16 + j]));
            mI = fmax(mI, fabs(Li[i * 16 + j]));
        }
    printf("{\"diag_block\": true, \"nb\": %d, \"fail\": %d, \"inv_err_rel\": %.2e, \"us_per_block\": %.3f, \"copy_us\": %.3f}\n", nb, f, eI / mI,
           (t[0] - t[1]) * 0.01 / reps, t[1] * 0.01 / reps);
    CK(hipFree(d_src)); CK(hipFree(d_I)); CK(hipFree(d_t)); CK(hipFree(d_f));
}

int main() {
    srand(7);
    for (int nb : {16, 15, 9, 1}) run_diag(nb);
    const int sizes[] = {1, 5, 15, 16, 17, 30, 31, 45, 60, 90, 150, 160, 165, 175};
    for (int n : sizes) {
        if (n <= 90) run<256>(n);
        run<512>(n);
    }
    return 0;
}
