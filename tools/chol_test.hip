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

template <int NT> __global__ __launch_bounds__(NT) void k_tiled(const double *src, int n, double *out, double *Lout, long long *ticks, int *failed) {
    extern __shared__ double lds[];
    __shared__ double Dblk[CH_NB][CH_NB + 1];
    __shared__ int fail;
    const int T = tl_tile_rows(n + 1);
    double *A = lds, *yv = A + tl_doubles(n + 1), *dinv = yv + 16 * T;
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
    const bool ok = tl_chol(A, n, n + 1, Dblk, dinv, &fail);
    __syncthreads();
    const long long t1 = wall_clock64();
    for (int i = threadIdx.x; i < 16 * T; i += NT) yv[i] = i < n ? A[tl_idx(n, i)] : 0.0;
    __syncthreads();
    if (ok) tl_trsv_t(A, n, dinv, yv);
    __syncthreads();
    const long long t2 = wall_clock64();
    for (int i = threadIdx.x; i < n; i += NT) out[i] = yv[i];
    for (int e = threadIdx.x; e < tri; e += NT) {
        int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > e) --i;
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        Lout[e] = A[tl_idx(i, e - i * (i + 1) / 2)];
    }
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = t2 - t1;
        ticks[2] = t0 - tc0;
        *failed = ok ? 0 : 1;
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
    long long tp[4] = {0}, tt[4] = {0};
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
        hipLaunchKernelGGL(k_tiled<NT>, dim3(1), dim3(NT), lds_t, 0, d_src, n, d_out, d_L, d_t, d_f);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(xt.data(), d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Lt.data(), d_L, sizeof(double) * tri, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tt, d_t, sizeof(long long) * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ft, d_f, sizeof(int), hipMemcpyDeviceToHost));
    }
    double xmax = 0, ep = 0, et = 0, el = 0, lmax = 0, ept = 0;
    for (int i = 0; i < n; ++i) xmax = fmax(xmax, fabs(x[i]));
    for (int i = 0; i < n; ++i) {
        ep = fmax(ep, fabs(xp[i] - x[i]));
        et = fmax(et, fabs(xt[i] - x[i]));
        ept = fmax(ept, fabs(xt[i] - xp[i]));
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            lmax = fmax(lmax, fabs(L[i * n + j]));
            el = fmax(el, fabs(Lt[i * (i + 1) / 2 + j] - L[i * n + j]));
        }
    printf("{\"threads\": %d, \"n\": %d, \"packed_fail\": %d, \"tiled_fail\": %d, \"x_err_packed\": %.2e, \"x_err_tiled\": %.2e, \"tiled_vs_packed\": %.2e, "
           "\"L_err_tiled\": %.2e, \"packed_us\": [%.2f, %.2f], \"tiled_us\": [%.2f, %.2f], \"tiled_copy_in_us\": %.2f}\n",
           NT, n, fp, ft, ep / xmax, et / xmax, ept / xmax, el / lmax, tp[0] * 0.01, tp[1] * 0.01, tt[0] * 0.01, tt[1] * 0.01, tt[2] * 0.01);
    hipFree(d_src); hipFree(d_out); hipFree(d_L); hipFree(d_t); hipFree(d_f);
}

int main() {
    srand(7);
    const int sizes[] = {1, 5, 15, 16, 17, 30, 31, 45, 60, 90, 150, 160, 165, 175};
    for (int n : sizes) {
        if (n <= 90) run<256>(n);
        run<512>(n);
    }
    return 0;
}
