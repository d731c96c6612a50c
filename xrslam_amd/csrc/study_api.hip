// study_api.hip -- BASELINE config 5's precision study ON THE DEVICE: the Schur contraction T = W^T diag(w) W of a window
// solve (the only GEMM-shaped piece of the path) with its operands in f64 (what the product computes,
// v_mfma_f64_16x16x4_f64), in f32 (v_mfma_f32_16x16x4_f32) and in bf16 with f32 accumulation (v_mfma_f32_16x16x16_bf16).
// Nothing of the product path calls this: the entry point times the three kernels on one operand and returns the products so
// that bench.py --workload s3 can report the error of the Gauss-Newton step each of them implies and the matrix-core rates.
// Same tiling as kb_schur_mfma (ba_kernels.hip.h): one 16x16 output tile per workgroup, the landmark range split over four
// wavefronts, partial tiles summed through LDS in a fixed order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/xrslam_hip.h"
#include "common.hip.h"

namespace xrhip {

typedef double study_d4 __attribute__((ext_vector_type(4)));
typedef float study_f4 __attribute__((ext_vector_type(4)));
typedef short study_s4 __attribute__((ext_vector_type(4)));

// A[l][a] = sqrt(w_l) W[l][a]: symmetric split, so that both MFMA operands carry the same rounding
template <int MODE>   // 0 = f64, 1 = f32, 2 = bf16 (A16: round-to-nearest-even upper halves of the f32 values)
__global__ __launch_bounds__(256) void ks_schur(const double *__restrict__ A64, const float *__restrict__ A32,
                                                const uint16_t *__restrict__ A16, int Lp, int PF, double *__restrict__ T) {
    __shared__ double red[4][256];
    const int tiles = PF / 16;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x - ti * tiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = Lp / 4, k0 = wave * kq;   // Lp is a multiple of 64: every wavefront's share is a multiple of 16
    const int i = lane & 15, kk = lane >> 4;
    double out[4];
    if (MODE == 0) {
        study_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k = k0; k < k0 + kq; k += 4) {
            const int l = k + kk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A64[(size_t)l * PF + 16 * ti + i], A64[(size_t)l * PF + 16 * tj + i], acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) out[r] = acc[r];
    } else if (MODE == 1) {
        study_f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k0 + kq; k += 4) {
            const int l = k + kk;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A32[(size_t)l * PF + 16 * ti + i], A32[(size_t)l * PF + 16 * tj + i], acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) out[r] = (double)acc[r];
    } else {
        study_f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k0 + kq; k += 16) {   // K = 16: lane (i, kk) carries landmarks k + 4 kk .. + 3
            study_s4 a, b;
            for (int q = 0; q < 4; ++q) {
                const int l = k + 4 * kk + q;
                a[q] = (short)A16[(size_t)l * PF + 16 * ti + i];
                b[q] = (short)A16[(size_t)l * PF + 16 * tj + i];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) out[r] = (double)acc[r];
    }
    // accumulator layouts: f32 / bf16 forms -- register r of lane (i, kk) is element (row 4 kk + r, column i); the f64 form
    // -- (row kk + 4 r, column i)
    for (int r = 0; r < 4; ++r) red[wave][(MODE == 0 ? kk + 4 * r : 4 * kk + r) * 16 + i] = out[r];
    __syncthreads();
    const int e = threadIdx.x;
    const double s = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    T[(size_t)(16 * ti + (e >> 4)) * PF + 16 * tj + (e & 15)] = s;
}

static uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    return (uint16_t)u;
}

}   // namespace xrhip

using namespace xrhip;

extern "C" int xrhip_study_schur_precision(const double *W, const double *w, int L, int P, int reps, double *out64, double *out32,
                                           double *out16, float ms_per_launch[3]) {
    if (!W || !w || !out64 || !out32 || !out16 || !ms_per_launch || L <= 0 || P <= 0 || reps <= 0)
        return xr_fail(XRHIP_EINVAL, "xrhip_study_schur_precision: bad arguments");
    const int PF = (P + 15) / 16 * 16, Lp = (L + 63) / 64 * 64;
    std::vector<double> a64((size_t)Lp * PF, 0.0);
    std::vector<float> a32((size_t)Lp * PF, 0.f);
    std::vector<uint16_t> a16((size_t)Lp * PF, 0);
    for (int l = 0; l < L; ++l) {
        if (!(w[l] >= 0.0)) return xr_fail(XRHIP_EINVAL, "xrhip_study_schur_precision: negative weight");
        const double s = std::sqrt(w[l]);
        for (int a = 0; a < P; ++a) {
            const double v = s * W[(size_t)l * P + a];
            a64[(size_t)l * PF + a] = v;
            a32[(size_t)l * PF + a] = (float)v;
            a16[(size_t)l * PF + a] = to_bf16((float)v);
        }
    }
    double *d64 = nullptr, *dT = nullptr;
    float *d32 = nullptr;
    uint16_t *d16 = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    XR_HIP(hipStreamCreate(&s));
    XR_HIP(hipEventCreate(&e0));
    XR_HIP(hipEventCreate(&e1));
    XR_HIP(hipMalloc(&d64, sizeof(double) * a64.size()));
    XR_HIP(hipMalloc(&d32, sizeof(float) * a32.size()));
    XR_HIP(hipMalloc(&d16, sizeof(uint16_t) * a16.size()));
    XR_HIP(hipMalloc(&dT, sizeof(double) * (size_t)PF * PF));
    XR_HIP(hipMemcpy(d64, a64.data(), sizeof(double) * a64.size(), hipMemcpyHostToDevice));
    XR_HIP(hipMemcpy(d32, a32.data(), sizeof(float) * a32.size(), hipMemcpyHostToDevice));
    XR_HIP(hipMemcpy(d16, a16.data(), sizeof(uint16_t) * a16.size(), hipMemcpyHostToDevice));
    const int grid = (PF / 16) * (PF / 16);
    std::vector<double> T((size_t)PF * PF);
    double *outs[3] = {out64, out32, out16};
    for (int mode = 0; mode < 3; ++mode) {
        for (int r = -3; r < reps; ++r) {   // three untimed launches first
            if (r == 0) XR_HIP(hipEventRecord(e0, s));
            if (mode == 0) hipLaunchKernelGGL(ks_schur<0>, dim3(grid), dim3(256), 0, s, d64, d32, d16, Lp, PF, dT);
            else if (mode == 1) hipLaunchKernelGGL(ks_schur<1>, dim3(grid), dim3(256), 0, s, d64, d32, d16, Lp, PF, dT);
            else hipLaunchKernelGGL(ks_schur<2>, dim3(grid), dim3(256), 0, s, d64, d32, d16, Lp, PF, dT);
        }
        XR_HIP(hipEventRecord(e1, s));
        XR_HIP(hipEventSynchronize(e1));
        XR_HIP(hipGetLastError());
        float ms = 0.f;
        XR_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_per_launch[mode] = ms / reps;
        XR_HIP(hipMemcpy(T.data(), dT, sizeof(double) * T.size(), hipMemcpyDeviceToHost));
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) outs[mode][(size_t)a * P + b] = T[(size_t)a * PF + b];
    }
    hipFree(d64);
    hipFree(d32);
    hipFree(d16);
    hipFree(dT);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipStreamDestroy(s);
    return XRHIP_OK;
}
