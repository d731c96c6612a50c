// Latency micro-benchmarks of ONE workgroup on an otherwise idle MI355X -- the regime of the single-workgroup BA kernels
// (kb_chain, kb_solve_try): dependent f64 chains, cross-lane broadcasts, LDS / L2 round trips, workgroup barriers,
// library transcendentals; and the shader clock such a kernel actually runs at (s_memtime vs the 100 MHz wall clock).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/latency.hip -o xrslam_amd/bin/xr-latency
// Prints one JSON line: ns per dependent operation (and shader cycles per operation).
// CAUTION (round 6): the one-operation chains below are loops of ONE operation per trip, so what they report is the operation PLUS the
// loop's compare-and-branch (~20 cycles): "fma_f64 32 cycles" is not a dependent latency.  tools/issue.hip measures the operations
// unrolled (v_fma_f64: 9 cycles, dependent or not); the multi-operation entries here (rsq_newton_pivot, mat3_product_scaled, the
// transcendental library calls, barriers and round trips) stand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Out {
  long long wall[32];   // 100 MHz ticks
  long long clk[32];    // shader cycles (s_memtime)
  double sink[32];
};

__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

#define BEGIN() do { __syncthreads(); w0 = wall_clock64(); c0 = __builtin_readcyclecounter(); } while (0)
#define END(slot, val) do { const long long c1 = __builtin_readcyclecounter(); const long long w1 = wall_clock64(); \
    if (threadIdx.x == 0) { o->wall[slot] = w1 - w0; o->clk[slot] = c1 - c0; o->sink[slot] = (val); } } while (0)

__global__ __launch_bounds__(256) void k_lat(Out *o, const double *g, double *gw, int N, double seed) {
  __shared__ double lds[1024];
  const int tid = threadIdx.x, lane = tid & 63;
  long long w0, c0;
  for (int i = tid; i < 1024; i += 256) lds[i] = seed + i * 1e-9;
  __syncthreads();
  double x = seed + tid * 1e-12, y = 1.0 + 1e-9 * tid;
  // 0: dependent v_fma_f64 chain
  BEGIN();
  for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
  END(0, x);
  // 1: dependent mul + add (contraction off: two instructions per step)
  BEGIN();
  for (int i = 0; i < N; ++i) x = x * y + 1e-9;
  END(1, x);
  // 2: readlane broadcast in the chain (value crosses lanes every step)
  BEGIN();
  for (int i = 0; i < N; ++i) x = bcast(x, i & 15) * y;
  END(2, x);
  // 3: ds_bpermute (shuffle) in the chain
  BEGIN();
  for (int i = 0; i < N; ++i) x = __shfl(x, (lane + 1) & 63) * y;
  END(3, x);
  // 4: LDS store -> load round trip in the chain (same wave, no barrier)
  BEGIN();
  for (int i = 0; i < N; ++i) {
    lds[tid] = x;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    x = lds[tid ^ 1] * y;
  }
  END(4, x);
  // 5: __syncthreads (4 waves) with an LDS exchange per step
  BEGIN();
  for (int i = 0; i < N; ++i) {
    lds[tid] = x;
    __syncthreads();
    x = lds[(tid + 64) & 255] * y;
    __syncthreads();
  }
  END(5, x);
  // 6: dependent global load (pointer chase through an L2-resident table of indices stored as doubles)
  {
    int idx = tid & 1023;
    BEGIN();
    for (int i = 0; i < N; ++i) idx = (int)g[idx];
    END(6, (double)idx);
  }
  // 7: f64 division chain
  BEGIN();
  for (int i = 0; i < N; ++i) x = 1.0 / (x + 1.5);
  END(7, x);
  // 8: f64 sqrt chain
  BEGIN();
  for (int i = 0; i < N; ++i) x = sqrt(x + 2.0);
  END(8, x);
  // 9: sin + cos chain (small arguments, like half rotation angles)
  BEGIN();
  for (int i = 0; i < N; ++i) x = 0.3 * sin(x) + 0.2 * cos(x);
  END(9, x);
  // 10: atan2 chain
  BEGIN();
  for (int i = 0; i < N; ++i) x = atan2(x + 0.1, 0.9);
  END(10, x);
  // 11: log chain (Cauchy loss)
  BEGIN();
  for (int i = 0; i < N; ++i) x = log(1.0 + x * x + 0.5);
  END(11, x);
  // 12: v_rsq_f64 + two coupled Newton steps (the Cholesky pivot)
  BEGIN();
  for (int i = 0; i < N; ++i) {
    const double d = x + 2.0, r0 = __builtin_amdgcn_rsq(d);
    double dd = d * r0, hh = 0.5 * r0, e = fma(-hh, dd, 0.5);
    dd = fma(dd, e, dd); hh = fma(hh, e, hh); e = fma(-hh, dd, 0.5);
    dd = fma(dd, e, dd); hh = fma(hh, e, hh);
    x = fma(fma(-dd, dd, d), hh, dd);
  }
  END(12, x);
  // 13: global store -> __threadfence -> barrier -> global load round trip (the orec hand-over inside kb_chain)
  BEGIN();
  for (int i = 0; i < N; ++i) {
    gw[tid] = x;
    __syncthreads();
    x = gw[(tid + 64) & 255] * y;
    __syncthreads();
  }
  END(13, x);
  // 14: independent f64 FMAs, 8 chains (issue rate rather than latency)
  {
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    BEGIN();
    for (int i = 0; i < N; ++i) {
      a0 = fma(a0, y, 1e-9); a1 = fma(a1, y, 1e-9); a2 = fma(a2, y, 1e-9); a3 = fma(a3, y, 1e-9);
      a4 = fma(a4, y, 1e-9); a5 = fma(a5, y, 1e-9); a6 = fma(a6, y, 1e-9); a7 = fma(a7, y, 1e-9);
    }
    END(14, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
  }
  // 16 / 17: independent v_mul_f64 / v_add_f64, 8 chains (issue rate of the single-pass operations)
  {
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    BEGIN();
    for (int i = 0; i < N; ++i) {
      a0 *= y; a1 *= y; a2 *= y; a3 *= y; a4 *= y; a5 *= y; a6 *= y; a7 *= y;
    }
    END(16, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    BEGIN();
    for (int i = 0; i < N; ++i) {
      a0 += y; a1 += y; a2 += y; a3 += y; a4 += y; a5 += y; a6 += y; a7 += y;
    }
    END(17, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
  }
  // 18: sincos() chain (one call for both)
  BEGIN();
  for (int i = 0; i < N; ++i) {
    double sv, cv;
    sincos(x, &sv, &cv);
    x = 0.3 * sv + 0.2 * cv;
  }
  END(18, x);
  // 19: one lane active (the others idle): dependent mul + add -- does the issue cost depend on the active lanes?
  BEGIN();
  if (lane == 0)
    for (int i = 0; i < N; ++i) x = x * y + 1e-9;
  END(19, x);
  // 20: 3x3 matrix product chain on one lane (45 instructions per product with contraction off)
  {
    double m[9], r[9];
    for (int i = 0; i < 9; ++i) m[i] = x * (i + 1) * 1e-3 + (i % 4 == 0 ? 1.0 : 0.0);
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = m[3 * i] * m[j] + m[3 * i + 1] * m[3 + j] + m[3 * i + 2] * m[6 + j];
#pragma unroll
      for (int i = 0; i < 9; ++i) m[i] = r[i] * 0.3 + (i % 4 == 0 ? 0.5 : 0.0);
    }
    END(20, m[0] + m[4] + m[8]);
  }
  // 15: wave-level butterfly sum of a double (6 shuffle stages)
  BEGIN();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    x *= 1e-2;
  }
  END(15, x);
}

int main() {
  Out *o;
  double *g, *gw;
  CK(hipMalloc(&o, sizeof(Out)));
  CK(hipMalloc(&g, 1024 * sizeof(double)));
  CK(hipMalloc(&gw, 1024 * sizeof(double)));
  double h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (double)((i * 389 + 17) & 1023);
  CK(hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice));
  CK(hipMemset(gw, 0, 1024 * sizeof(double)));
  const int N = 2000;
  Out r;
  for (int rep = 0; rep < 3; ++rep) {   // the last repetition is reported (warm caches, settled clocks)
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(256), 0, 0, o, g, gw, N, 0.37);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&r, o, sizeof(r), hipMemcpyDeviceToHost));
  }
  const char *names[21] = {"fma_f64", "mul_add_f64", "readlane_bcast_mul", "bpermute_mul", "lds_roundtrip_mul", "syncthreads_x2_lds",
                           "global_load_chase", "div_f64", "sqrt_f64", "sin_plus_cos", "atan2", "log", "rsq_newton_pivot",
                           "global_store_barrier_load", "fma_f64_8_chains_per_8", "wave_sum_f64", "mul_f64_8_chains_per_8",
                           "add_f64_8_chains_per_8", "sincos_one_call", "mul_add_one_lane", "mat3_product_scaled"};
  printf("{\"n\": %d", N);
  for (int s = 0; s < 21; ++s)
    printf(", \"%s\": {\"ns\": %.2f, \"cycles\": %.1f}", names[s], r.wall[s] * 10.0 / N, (double)r.clk[s] / N);
  double ghz = 0;
  for (int s = 0; s < 16; ++s) ghz += (double)r.clk[s] / (r.wall[s] * 10.0);
  printf(", \"shader_clock_ghz_single_workgroup\": %.3f}\n", ghz / 16);
  return 0;
}
